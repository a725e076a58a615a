"""Tile decode (image-stitcher_amd/tiff.py): the TIFF container read against Pillow-written files and
hand-built ones, and the reference's page selection rules (registration/tile_registration.py:1423-1447,
:2184-2279; image_loaders.py:141-347).  tifffile, which the reference decodes with, is not importable
here: the container parity is pinned by Pillow (an independent writer and reader), the selection rules
by their statement in the reference."""
import os
import struct

import numpy as np
import pytest

from image_stitcher_amd import tiff

PIL = pytest.importorskip("PIL.Image")


def _img(seed, h, w):
    return np.random.default_rng(seed).integers(0, 65536, size=(h, w), dtype=np.uint16)


def _save(path, arrays, **kw):
    ims = [PIL.fromarray(a) for a in arrays]
    ims[0].save(path, format="TIFF", save_all=len(ims) > 1, append_images=ims[1:], **kw)


def test_single_page_uncompressed(tmp_path):
    a = _img(1, 300, 500)          # several strips
    p = str(tmp_path / "R_3_0_Fluorescence_488_nm_Ex.tiff")
    _save(p, [a])
    t = tiff.TiffFile(p)
    assert len(t.pages) == 1 and t.pages[0].raw and len(t.pages[0].strip_offsets) >= 1
    assert np.array_equal(t.read_page(0), a)
    assert np.array_equal(tiff.read_tile(p), a)
    # in place, into rows of a larger buffer (what the pinned staging buffer is)
    stack = np.zeros((3, 300, 500), dtype=np.uint16)
    t.read_page_into(0, stack[1])
    assert np.array_equal(stack[1], a) and not stack[0].any() and not stack[2].any()
    with pytest.raises(ValueError):
        t.read_page_into(0, np.zeros((10, 10), dtype=np.uint16))


def test_eight_bit_and_compressed_pages(tmp_path):
    a8 = np.random.default_rng(2).integers(0, 256, size=(64, 96), dtype=np.uint8)
    p8 = str(tmp_path / "A_0_0_BF.tiff")
    _save(p8, [a8])
    assert np.array_equal(tiff.read_tile(p8), a8)
    a = _img(3, 128, 200)
    for comp in ("tiff_lzw", "tiff_adobe_deflate"):
        p = str(tmp_path / f"A_1_0_{comp}.tiff")
        _save(p, [a], compression=comp)
        t = tiff.TiffFile(p)
        assert not t.pages[0].raw
        assert np.array_equal(t.read_page(0), a)       # through Pillow, as the reference falls back to it


def _handmade(path, a, big_endian=False, bigtiff=False, strips=3, byte_counts=True, drop_last_strip=False,
              sample_format=None, entry_count=None):
    """A minimal TIFF written field by field: chosen byte order, classic or BigTIFF, several strips.
    ``byte_counts=False`` leaves StripByteCounts (tag 279) out, ``drop_last_strip`` lists one strip offset too few,
    ``sample_format`` adds tag 339, ``entry_count`` overrides the directory's entry count (a corrupt file)."""
    bo = ">" if big_endian else "<"
    h, w = a.shape
    rps = (h + strips - 1) // strips
    data = a.astype(a.dtype.newbyteorder(bo)).tobytes()
    row = w * a.itemsize
    offs, cnts = [], []
    head = 16 if bigtiff else 8
    pos = head
    for r0 in range(0, h, rps):
        n = (min(h, r0 + rps) - r0) * row
        offs.append(pos)
        cnts.append(n)
        pos += n
    n_strips = len(offs)
    otype, osize, ofmt = (16, 8, "Q") if bigtiff else (4, 4, "I")
    arrays = b"".join(struct.pack(bo + ofmt, v) for v in offs) + b"".join(struct.pack(bo + ofmt, v) for v in cnts)
    offs_at, cnts_at = pos, pos + n_strips * osize
    ifd_at = pos + len(arrays)
    inline = 8 if bigtiff else 4

    def entry(tag, typ, count, value):
        fmt = {3: "H", 4: "I", 16: "Q"}[typ]
        if isinstance(value, bytes):
            payload = value
        else:
            payload = struct.pack(bo + fmt, value)
        payload = payload + b"\0" * (inline - len(payload))
        return struct.pack(bo + "HH", tag, typ) + struct.pack(bo + ("Q" if bigtiff else "I"), count) + payload

    ptr = lambda v: struct.pack(bo + ("Q" if bigtiff else "I"), v)
    one_strip = n_strips == 1
    listed = n_strips - 1 if drop_last_strip else n_strips
    entries = [entry(256, 4, 1, w), entry(257, 4, 1, h), entry(258, 3, 1, a.itemsize * 8), entry(259, 3, 1, 1),
               entry(262, 3, 1, 1),
               entry(273, otype, listed, offs[0] if listed == 1 else ptr(offs_at)),
               entry(277, 3, 1, 1), entry(278, 4, 1, rps)]
    if byte_counts:
        entries.append(entry(279, otype, listed, cnts[0] if listed == 1 else ptr(cnts_at)))
    if sample_format is not None:
        entries.append(entry(339, 3, 1, sample_format))
    n_entries = len(entries) if entry_count is None else entry_count
    if bigtiff:
        header = (b"MM" if big_endian else b"II") + struct.pack(bo + "HHH", 43, 8, 0) + struct.pack(bo + "Q", ifd_at)
        ifd = struct.pack(bo + "Q", n_entries) + b"".join(entries) + struct.pack(bo + "Q", 0)
    else:
        header = (b"MM" if big_endian else b"II") + struct.pack(bo + "H", 42) + struct.pack(bo + "I", ifd_at)
        ifd = struct.pack(bo + "H", n_entries) + b"".join(entries) + struct.pack(bo + "I", 0)
    with open(path, "wb") as f:
        f.write(header + data + arrays + ifd)


@pytest.mark.parametrize("big_endian,bigtiff,strips", [(False, False, 3), (True, False, 3), (False, True, 4), (True, True, 1),
                                                        (False, False, 1)])
def test_byte_orders_and_bigtiff(tmp_path, big_endian, bigtiff, strips):
    a = _img(4, 77, 130)
    p = str(tmp_path / "B_0_0_x.tiff")
    _handmade(p, a, big_endian, bigtiff, strips)
    t = tiff.TiffFile(p)
    assert t.pages[0].raw and len(t.pages[0].strip_offsets) == strips
    assert np.array_equal(t.read_page(0), a)
    if not bigtiff:                                    # Pillow reads classic TIFFs of either byte order: same pixels
        with PIL.open(p) as im:
            assert np.array_equal(np.asarray(im), a)


@pytest.mark.parametrize("strips", [1, 3])
def test_a_file_without_strip_byte_counts_is_read_whole(tmp_path, strips):
    """Old writers leave StripByteCounts out: the counts follow from the rows of every strip."""
    a = _img(5, 60, 90)
    p = str(tmp_path / "C_0_0_x.tiff")
    _handmade(p, a, strips=strips, byte_counts=False)
    t = tiff.TiffFile(p)
    assert t.pages[0].strip_counts == []
    dst = np.full((60, 90), 0xABCD, dtype=np.uint16)      # stale contents of a reused staging buffer
    t.read_page_into(0, dst)
    assert np.array_equal(dst, a)


def test_pages_that_cannot_fill_the_destination_are_refused(tmp_path):
    a = _img(6, 60, 90)
    p = str(tmp_path / "D_0_0_x.tiff")
    _handmade(p, a, strips=3, drop_last_strip=True)        # a strip is missing: part of the page would stay stale
    with pytest.raises(ValueError, match="strip"):
        tiff.TiffFile(p).read_page_into(0, np.zeros((60, 90), dtype=np.uint16))
    for fmt, dtype in ((2, np.int16), (3, np.float16)):    # signed / floating-point samples of the same item size
        _handmade(p, a.view(dtype), strips=2, sample_format=fmt)
        t = tiff.TiffFile(p)
        with pytest.raises(ValueError, match="format"):
            t.read_page_into(0, np.zeros((60, 90), dtype=np.uint16))
        assert t.read_page(0).dtype == np.dtype(dtype)     # into a buffer of its own type it still reads
    for big in (False, True):                              # an entry count far beyond the file
        _handmade(p, a, bigtiff=big, entry_count=60000 if not big else 1 << 40)
        with pytest.raises(ValueError, match="entries"):
            tiff.TiffFile(p)


def test_stack_files_give_their_middle_page(tmp_path):
    pages = [_img(10 + i, 40, 60) for i in range(5)]
    p = str(tmp_path / "C_7_stack.tiff")
    _save(p, pages)
    t = tiff.TiffFile(p)
    assert len(t.pages) == 5
    assert tiff.select_page(t) == 2
    assert np.array_equal(tiff.read_tile(p), pages[2])
    # a regular name takes the file's first image whatever else it holds
    q = str(tmp_path / "C_7_0_x.tiff")
    _save(q, pages)
    assert tiff.select_page(tiff.TiffFile(q)) == 0
    # four pages: index 4 // 2
    r = str(tmp_path / "C_8_stack.tiff")
    _save(r, pages[:4])
    assert tiff.select_page(tiff.TiffFile(r)) == 2


def _ome_xml(c, z, t, order):
    chans = "".join(f'<Channel ID="Channel:0:{i}" Name="ch{i}" SamplesPerPixel="1"/>' for i in range(c))
    return ('<?xml version="1.0" encoding="UTF-8"?><OME xmlns="http://www.openmicroscopy.org/Schemas/OME/2016-06">'
            f'<Image ID="Image:0"><Pixels ID="Pixels:0" DimensionOrder="{order}" Type="uint16" SizeX="60" SizeY="40" '
            f'SizeC="{c}" SizeZ="{z}" SizeT="{t}">{chans}<TiffData/></Pixels></Image></OME>')


@pytest.mark.parametrize("order", ["XYZCT", "XYCZT"])
def test_ome_tiff_page_of_channel_and_z(tmp_path, order):
    C, Z = 3, 5
    pages = [_img(100 + i, 40, 60) for i in range(C * Z)]
    p = str(tmp_path / "D_4.ome.tiff")
    _save(p, pages, description=_ome_xml(C, Z, 1, order))
    t = tiff.TiffFile(p)
    dims = t.ome_dims()
    assert dims["C"] == C and dims["Z"] == Z and dims["order"] == order and dims["channel_names"] == ["ch0", "ch1", "ch2"]
    plane = (lambda c, z: c * Z + z) if order == "XYZCT" else (lambda c, z: z * C + c)
    for c in range(C):
        # middle slice num_z // 2 by default (tile_registration.py:2220)
        assert tiff.select_page(t, channel=c) == plane(c, Z // 2)
        assert np.array_equal(tiff.read_tile(p, channel=c), pages[plane(c, 2)])
        # the per-FOV override wins (:2223-2228); other FOVs' entries do not apply
        assert tiff.select_page(t, {4: 0}, channel=c) == plane(c, 0)
        assert tiff.select_page(t, {5: 0}, channel=c) == plane(c, 2)
    with pytest.raises(ValueError):
        t.ome_page(channel=C, z=0)
    with pytest.raises(ValueError):
        t.ome_page(channel=0, z=Z)
    # an .ome.tiff without OME-XML: the reference's tifffile fallback takes the middle page (:2236-2247)
    q = str(tmp_path / "D_5.ome.tiff")
    _save(q, pages[:4])
    assert tiff.select_page(tiff.TiffFile(q)) == 2


def test_ome_tiffdata_blocks_map_planes_to_pages(tmp_path):
    """TiffData elements that reorder and offset the planes (OME-TIFF specification); hand-built, tifffile absent."""
    C, Z = 2, 3
    pages = [_img(300 + i, 40, 60) for i in range(C * Z + 1)]       # page 0 is a stray leading page
    chans = "".join(f'<Channel ID="Channel:0:{i}" Name="ch{i}"/>' for i in range(C))
    head = ('<?xml version="1.0" encoding="UTF-8"?><OME xmlns="http://www.openmicroscopy.org/Schemas/OME/2016-06">'
            f'<Image ID="Image:0"><Pixels ID="Pixels:0" DimensionOrder="XYZCT" Type="uint16" SizeX="60" SizeY="40" '
            f'SizeC="{C}" SizeZ="{Z}" SizeT="1">{chans}')
    tail = '</Pixels></Image></OME>'
    # channel 1 first (pages 1..3), then channel 0 (pages 4..6), one block of three planes each
    blocks = '<TiffData IFD="1" FirstC="1" FirstZ="0" PlaneCount="3"/><TiffData IFD="4" FirstC="0" FirstZ="0" PlaneCount="3"/>'
    p = str(tmp_path / "E_1.ome.tiff")
    _save(p, pages, description=head + blocks + tail)
    t = tiff.TiffFile(p)
    assert [t.ome_page(channel=1, z=z) for z in range(Z)] == [1, 2, 3]
    assert [t.ome_page(channel=0, z=z) for z in range(Z)] == [4, 5, 6]
    assert np.array_equal(tiff.read_tile(p, channel=0), pages[5])                  # the middle z of channel 0
    # one TiffData per plane with IFD only counts one plane; a plane without a block raises
    single = "".join(f'<TiffData IFD="{6 - k}" FirstC="{k // Z}" FirstZ="{k % Z}"/>' for k in range(C * Z - 1))
    q = str(tmp_path / "E_2.ome.tiff")
    _save(q, pages, description=head + single + tail)
    tq = tiff.TiffFile(q)
    assert tq.ome_page(channel=0, z=0) == 6 and tq.ome_page(channel=1, z=1) == 2
    with pytest.raises(ValueError, match="no TiffData block"):
        tq.ome_page(channel=1, z=2)
    # a plane that lives in another file of a multi-file set
    multi = ('<TiffData IFD="0" FirstC="0" PlaneCount="3"><UUID FileName="E_3.ome.tiff">urn:uuid:1</UUID></TiffData>'
             '<TiffData IFD="0" FirstC="1" PlaneCount="3"><UUID FileName="E_3_c1.ome.tiff">urn:uuid:2</UUID></TiffData>')
    r = str(tmp_path / "E_3.ome.tiff")
    _save(r, pages, description=head + multi + tail)
    tr_ = tiff.TiffFile(r)
    assert tr_.ome_page(channel=0, z=2) == 2
    with pytest.raises(ValueError, match="E_3_c1.ome.tiff"):
        tr_.ome_page(channel=1, z=0)
    # sizes that are not numbers: no usable OME description, the reference's middle-page fallback (:2236-2247)
    bad = str(tmp_path / "E_4.ome.tiff")
    _save(bad, pages[:5], description=head.replace(f'SizeZ="{Z}"', 'SizeZ="many"') + tail)
    assert tiff.TiffFile(bad).ome_dims() is None and tiff.select_page(tiff.TiffFile(bad)) == 2


def test_filename_patterns():
    assert tiff.parse_filename_fov_info("R0_12.ome.tiff") == {"region": "R0", "fov": 12}
    assert tiff.parse_filename_fov_info("R0_12_3_Fluorescence_488_nm_Ex.tiff") == {"region": "R0", "fov": 12, "z_level": 3}
    assert tiff.parse_filename_fov_info("manual_5_stack.tiff") == {"region": "manual", "fov": 5}
    assert tiff.parse_filename_fov_info("notes.txt") is None


def test_middle_z_levels_rule():
    # five z levels per FOV, rows shuffled; FOV 2 has four (index 4 // 2 = 2 -> third smallest)
    fov = np.array([0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2])
    z = np.array([4, 0, 1, 3, 0, 1, 3, 4, 2, 2, 7, 5, 6, 4])
    got = tiff.middle_z_levels(fov, z)
    assert got == {0: 2, 1: 2, 2: 6}
    assert tiff.middle_z_levels(fov, z, fovs_in_use=[2, 9]) == {2: 6}
    pd = pytest.importorskip("pandas")
    # the reference's own statement of the rule (tile_registration.py:1440-1444) on the same table
    df = pd.DataFrame({"fov": fov, "z_level": z})
    want = {}
    for f in sorted(set(fov.tolist())):
        rows = df[df["fov"].astype(int) == f].sort_values("z_level")
        want[f] = int(rows.iloc[len(rows) // 2]["z_level"])
    assert got == want


def test_fill_from_tiffs(tmp_path):
    tiles = [_img(200 + i, 64, 96) for i in range(7)]
    paths = []
    for i, a in enumerate(tiles):
        if i == 3:      # one stack file among regular ones: its middle page is the tile
            p = str(tmp_path / f"E_{i}_stack.tiff")
            _save(p, [_img(900, 64, 96), a, _img(901, 64, 96)])
        else:
            p = str(tmp_path / f"E_{i}_0_ch.tiff")
            _save(p, [a])
        paths.append(p)
    for workers in (1, 4):
        fill = tiff.fill_from_tiffs(paths, workers=workers)
        dst = np.zeros((4, 64, 96), dtype=np.uint16)
        fill(2, 6, dst)
        assert np.array_equal(dst, np.stack(tiles[2:6]))


@pytest.mark.gpu
def test_register_streaming_from_tiff_files(tmp_path):
    """Tiles decoded from files into the pinned buffers and registered band by band == the same tiles resident."""
    import torch
    from image_stitcher_amd import ingest, registration as reg, synth
    g = synth.make_grid(5, 3, 4, 128, 128, overlap=0.15, jitter=3)
    paths = []
    for i, t in enumerate(g.tiles):
        p = str(tmp_path / f"F_{i}_0_ch.tiff")
        _save(p, [t])
        paths.append(p)
    tiles_dev, out, status = ingest.register_streaming(tiff.fill_from_tiffs(paths), 3, 4, 128, 128, g.pair_idx, g.pair_dir,
                                                       edge_width=32, rows_per_band=1)
    ref_out, ref_status = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 32)
    assert np.array_equal(tiles_dev.cpu().view(torch.int16).numpy().view(np.uint16), g.tiles)
    assert torch.equal(out, ref_out) and torch.equal(status, ref_status)
