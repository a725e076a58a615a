"""Tile decode for the ingest stage: TIFF pages straight into pinned staging buffers, and the
reference's rules for which page of a file is the tile.

The reference loads every tile through tifffile (``load_single_image``,
registration/tile_registration.py:2184-2279; loaders in image_loaders.py:141-347) and picks

  *  ``*.ome.tif[f]``   page of (channel, z), z = the middle slice ``num_z // 2`` unless the
                        per-FOV override ``fov_to_z_level`` names another (:2223-2234)
  *  ``*_stack*``       the middle page ``num_pages // 2`` (:2253-2262)
  *  anything else      the file's image (:2264-2266)

where the override comes from the coordinates table: per field of view the rows sorted by
``z_level``, the one at index ``len // 2`` (:1423-1447).

Microscope tiles are uncompressed baseline TIFFs (one or more strips of raw little-endian
samples), so "decoding" is locating the strips: ``TiffFile`` parses the IFD chain (classic
and BigTIFF, either byte order) and ``read_page_into`` reads the strips directly into the
destination rows, which for the ingest stage is the pinned buffer the H2D copy starts from
(ingest.TileStager) - no intermediate array.  Compressed or tiled pages are handed to Pillow
when it is importable (the reference's own fallback, image_loaders.py:205-208) and refused
otherwise.  Host code only: the device side of ingest is the upload in ingest.py.
"""
from __future__ import annotations

import os
import re
import struct
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# the three filename shapes of the reference (tile_registration.py:100-104)
DEFAULT_FOV_RE = re.compile(r"(?P<region>\w+)_(?P<fov>0|[1-9]\d*)_(?P<z_level>\d+)_", re.I)
MULTIPAGE_FOV_RE = re.compile(r"(?P<region>\w+)_(?P<fov>0|[1-9]\d*)_stack", re.I)
OME_TIFF_FOV_RE = re.compile(r"(?P<region>\w+)_(?P<fov>0|[1-9]\d*)\.ome\.tiff?$", re.I)


def parse_filename_fov_info(filename: str) -> Optional[Dict[str, object]]:
    """region / fov (/ z_level) of a tile file name, patterns tried in the reference's order
    (tile_registration.py:131-174): OME-TIFF, regular, multi-page."""
    m = OME_TIFF_FOV_RE.search(filename)
    if m:
        return {"region": m.group("region"), "fov": int(m.group("fov"))}
    m = DEFAULT_FOV_RE.search(filename)
    if m:
        return {"region": m.group("region"), "fov": int(m.group("fov")), "z_level": int(m.group("z_level"))}
    m = MULTIPAGE_FOV_RE.search(filename)
    if m:
        return {"region": m.group("region"), "fov": int(m.group("fov"))}
    return None


def is_ome_tiff_name(name: str) -> bool:
    low = name.lower()
    return low.endswith(".ome.tif") or low.endswith(".ome.tiff")


def middle_z_levels(fov: Sequence[int], z_level: Sequence[int], fovs_in_use: Optional[Sequence[int]] = None) -> Dict[int, int]:
    """``fov_to_z_level`` of tile_registration.py:1423-1447: for every field of view in use, the
    rows of the coordinates table sorted by z_level (a stable sort, as pandas' default is not
    relied upon: equal z_levels do not change the answer), the z_level at index ``len // 2``."""
    fov = np.asarray(fov).astype(np.int64)
    z = np.asarray(z_level)
    wanted = sorted(set(int(f) for f in (fov if fovs_in_use is None else fovs_in_use)))
    out: Dict[int, int] = {}
    for f in wanted:
        rows = np.sort(z[fov == f], kind="stable")
        if len(rows):
            out[f] = int(rows[len(rows) // 2])
    return out


# ---- the container ---------------------------------------------------------------------------------
_TYPE_SIZE = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8, 17: 8, 18: 8}
_TYPE_FMT = {1: "B", 3: "H", 4: "I", 6: "b", 8: "h", 9: "i", 16: "Q", 17: "q", 18: "Q"}


@dataclass
class TiffPage:
    width: int
    height: int
    bits: int
    samples: int
    sample_format: int          # 1 unsigned, 2 signed, 3 float
    compression: int
    rows_per_strip: int
    strip_offsets: List[int]
    strip_counts: List[int]
    tiled: bool
    description: Optional[str] = None
    byteorder: str = "<"

    @property
    def dtype(self) -> np.dtype:
        kind = {1: "u", 2: "i", 3: "f"}.get(self.sample_format, "u")
        return np.dtype(f"{self.byteorder}{kind}{self.bits // 8}")

    @property
    def raw(self) -> bool:
        """Strips of raw samples that can be read in place."""
        return self.compression == 1 and not self.tiled and self.samples == 1 and self.bits in (8, 16, 32)


@dataclass
class TiffFile:
    """The IFD chain of a TIFF or BigTIFF file (top-level pages only; SubIFDs are not followed)."""
    path: str
    pages: List[TiffPage] = field(default_factory=list)

    def __post_init__(self):
        with open(self.path, "rb") as f:
            head = f.read(16)
            if len(head) < 8 or head[:2] not in (b"II", b"MM"):
                raise ValueError(f"{self.path}: not a TIFF file")
            bo = "<" if head[:2] == b"II" else ">"
            magic = struct.unpack(bo + "H", head[2:4])[0]
            if magic == 42:
                big, offset = False, struct.unpack(bo + "I", head[4:8])[0]
            elif magic == 43:
                big, offset = True, struct.unpack(bo + "Q", head[8:16])[0]
            else:
                raise ValueError(f"{self.path}: unknown TIFF version {magic}")
            size = os.fstat(f.fileno()).st_size
            seen = set()
            while offset and offset not in seen:
                seen.add(offset)
                if offset >= size:
                    raise ValueError(f"{self.path}: IFD offset {offset} beyond the end of the file")
                offset = self._read_ifd(f, bo, big, offset, size)

    def _read_ifd(self, f, bo: str, big: bool, offset: int, size: int) -> int:
        f.seek(offset)
        n = struct.unpack(bo + ("Q" if big else "H"), f.read(8 if big else 2))[0]
        esz, inline = (20, 8) if big else (12, 4)
        # the entry count comes from the file: a directory cannot be longer than what is left of the file
        if offset + (8 if big else 2) + n * esz + (8 if big else 4) > size:
            raise ValueError(f"{self.path}: IFD at {offset} claims {n} entries, beyond the end of the file")
        raw = f.read(n * esz + (8 if big else 4))
        tags: Dict[int, Tuple[int, int, bytes]] = {}
        for i in range(n):
            e = raw[i * esz:(i + 1) * esz]
            tag, typ = struct.unpack(bo + "HH", e[:4])
            count = struct.unpack(bo + ("Q" if big else "I"), e[4:4 + inline])[0]
            tags[tag] = (typ, count, e[4 + inline:])
        nxt = struct.unpack(bo + ("Q" if big else "I"), raw[n * esz:])[0]

        def values(tag, default=None):
            if tag not in tags:
                return default
            typ, count, payload = tags[tag]
            nbytes = _TYPE_SIZE.get(typ, 1) * count
            if nbytes > inline:
                at = struct.unpack(bo + ("Q" if big else "I"), payload[:inline])[0]
                if at + nbytes > size:
                    raise ValueError(f"{self.path}: tag {tag} points beyond the end of the file")
                f.seek(at)
                data = f.read(nbytes)
            else:
                data = payload[:nbytes]
            if typ == 2:
                return data.split(b"\0", 1)[0].decode("utf-8", "replace")
            if typ in _TYPE_FMT:
                return list(struct.unpack(bo + _TYPE_FMT[typ] * count, data))
            return data

        def one(tag, default):
            v = values(tag)
            return default if v is None else (v[0] if isinstance(v, list) else v)

        bits = values(258, [1])
        samples = one(277, 1)
        height = one(257, 0)
        tiled = 322 in tags or 324 in tags
        rps = min(one(278, height) or height, height) if height else 0
        self.pages.append(TiffPage(
            width=one(256, 0), height=height, bits=int(bits[0]), samples=int(samples), sample_format=int(one(339, 1)),
            compression=int(one(259, 1)), rows_per_strip=int(rps), strip_offsets=[int(v) for v in (values(273) or [])],
            strip_counts=[int(v) for v in (values(279) or [])], tiled=tiled,
            description=values(270) if 270 in tags and tags[270][0] == 2 else None, byteorder=bo))
        return nxt

    # -- OME ---------------------------------------------------------------------------------------
    def _ome_pixels(self):
        """The first Pixels element of the OME-XML in the first page's description, or None."""
        desc = self.pages[0].description if self.pages else None
        if not desc or "<OME" not in desc:
            return None
        try:
            root = ET.fromstring(desc)
        except ET.ParseError:
            return None
        for el in root.iter():
            if el.tag.endswith("}Pixels") or el.tag == "Pixels":
                return el
        return None

    def ome_dims(self) -> Optional[Dict[str, object]]:
        """SizeC / SizeZ / SizeT and DimensionOrder of the first Image of the OME-XML in the first page's
        description; None when the file carries none, or sizes that are not positive numbers (the caller then
        takes the reference's fallback for an OME loader that fails, tile_registration.py:2236-2247)."""
        el = self._ome_pixels()
        if el is None:
            return None
        names = [c.get("Name") or c.get("ID") for c in el if c.tag.endswith("Channel")]
        try:
            dims = {"C": int(el.get("SizeC", 1)), "Z": int(el.get("SizeZ", 1)), "T": int(el.get("SizeT", 1))}
        except (TypeError, ValueError):
            return None
        order = el.get("DimensionOrder", "XYZCT")
        if min(dims.values()) < 1 or len(order) != 5 or order[:2] != "XY" or sorted(order[2:]) != ["C", "T", "Z"]:
            return None
        return dict(dims, order=order, channel_names=names or None)

    def ome_page(self, channel: int = 0, z: int = 0, t: int = 0) -> int:
        """Index of the page (IFD) that holds (t, channel, z).  The planes are numbered in DimensionOrder, its first
        letter after XY varying fastest; the TiffData elements of the Pixels element say which IFDs hold which planes
        (OME-TIFF specification: a block starts at plane (FirstC, FirstZ, FirstT), default 0, covers PlaneCount planes,
        default 1 when IFD is given and all planes otherwise, and sits at IFD, IFD + 1, ...; default IFD 0).  Without
        TiffData elements plane k is IFD k.  This is what ``OMETiffLoader.read_slice`` reaches through tifffile's
        series axes (image_loaders.py:270-316); tifffile is not importable here, so files whose TiffData blocks reorder
        planes are covered by hand-built cases only (parity with tifffile unpinned).  A plane that a block places in
        ANOTHER file of a multi-file set (UUID FileName) raises."""
        dims = self.ome_dims()
        if dims is None:
            raise ValueError(f"{self.path}: no OME-XML")
        for name, v in (("channel", (channel, dims["C"])), ("z", (z, dims["Z"])), ("t", (t, dims["T"]))):
            if not 0 <= v[0] < v[1]:
                raise ValueError(f"{name} {v[0]} out of range [0, {v[1]})")

        def plane_number(c_, z_, t_):
            index, stride = 0, 1
            at = {"C": c_, "Z": z_, "T": t_}
            for axis in str(dims["order"])[2:]:
                index += at[axis] * stride
                stride *= int(dims[axis])
            return index

        index = plane_number(channel, z, t)
        total = dims["C"] * dims["Z"] * dims["T"]
        blocks = [c for c in self._ome_pixels() if c.tag.endswith("TiffData")]
        if not blocks:
            return index

        def num(el, key, default):
            try:
                return int(el.get(key, default))
            except (TypeError, ValueError):
                raise ValueError(f"{self.path}: TiffData attribute {key}={el.get(key)!r} is not a number")

        for td in blocks:
            start = plane_number(num(td, "FirstC", 0), num(td, "FirstZ", 0), num(td, "FirstT", 0))
            count = num(td, "PlaneCount", 1 if td.get("IFD") is not None else total - start)
            if start <= index < start + count:
                for child in td:
                    other = child.get("FileName") if child.tag.endswith("UUID") else None
                    if other and os.path.basename(other) != os.path.basename(self.path):
                        raise ValueError(f"{self.path}: plane (c={channel}, z={z}, t={t}) is stored in {other}")
                ifd = num(td, "IFD", 0) + (index - start)
                if not 0 <= ifd < len(self.pages):
                    raise ValueError(f"{self.path}: TiffData points at IFD {ifd} of {len(self.pages)}")
                return ifd
        raise ValueError(f"{self.path}: no TiffData block holds plane (c={channel}, z={z}, t={t})")

    # -- pixels ------------------------------------------------------------------------------------
    def read_page_into(self, page_index: int, dst: np.ndarray) -> None:
        """The page's pixels into ``dst`` ([H, W], C-contiguous rows, native byte order, same item size)."""
        page = self.pages[page_index]
        if dst.shape != (page.height, page.width):
            raise ValueError(f"{self.path} page {page_index} is {page.height} x {page.width}, the destination {dst.shape}")
        if not page.raw:
            _pillow_read(self.path, page_index, dst)
            return
        if dst.dtype.itemsize != page.bits // 8 or not dst.flags.c_contiguous:
            raise ValueError(f"destination must be C-contiguous with {page.bits // 8}-byte items")
        # same item size is not the same type: signed or floating-point samples are not copied into unsigned pixels
        want_kind = {1: "u", 2: "i", 3: "f"}.get(page.sample_format)
        if want_kind is None or dst.dtype.kind != want_kind:
            raise ValueError(f"{self.path} page {page_index}: samples of format {page.sample_format} "
                             f"({page.dtype}) do not go into a {dst.dtype} destination")
        row_bytes = page.width * page.bits // 8
        # every row has to come from a strip: the destination is uninitialised (or reused) memory
        n_strips = -(-page.height // page.rows_per_strip) if page.rows_per_strip else 0
        if len(page.strip_offsets) != n_strips:
            raise ValueError(f"{self.path} page {page_index}: {len(page.strip_offsets)} strip offsets for {n_strips} strips "
                             f"({page.height} rows, {page.rows_per_strip} per strip)")
        counts = page.strip_counts
        if not counts:
            # StripByteCounts (tag 279) absent, as old single-strip writers leave it: uncompressed strips hold whole rows
            counts = [(min(page.height, (k + 1) * page.rows_per_strip) - k * page.rows_per_strip) * row_bytes
                      for k in range(n_strips)]
        elif len(counts) != n_strips:
            raise ValueError(f"{self.path} page {page_index}: {len(counts)} strip byte counts for {n_strips} strips")
        flat = dst.reshape(-1).view(np.uint8)
        with open(self.path, "rb", buffering=0) as f:
            for s, (off, cnt) in enumerate(zip(page.strip_offsets, counts)):
                r0 = s * page.rows_per_strip
                want = (min(page.height, r0 + page.rows_per_strip) - r0) * row_bytes
                if cnt < want:
                    raise ValueError(f"{self.path} page {page_index}: strip {s} holds {cnt} bytes, needs {want}")
                f.seek(off)
                view = memoryview(flat[r0 * row_bytes:r0 * row_bytes + want])
                got = 0
                while got < want:
                    k = f.readinto(view[got:])
                    if not k:
                        raise ValueError(f"{self.path}: truncated strip {s} of page {page_index}")
                    got += k
        if page.bits > 8 and page.byteorder != ("<" if np.little_endian else ">"):
            dst.byteswap(inplace=True)

    def read_page(self, page_index: int) -> np.ndarray:
        page = self.pages[page_index]
        out = np.empty((page.height, page.width), dtype=page.dtype.newbyteorder("="))
        self.read_page_into(page_index, out)
        return out


def _pillow_read(path: str, page_index: int, dst: np.ndarray) -> None:
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover - depends on the environment
        raise ValueError(f"{path}: compressed or tiled TIFF pages need Pillow, which is not importable") from e
    with Image.open(path) as img:
        img.seek(page_index)
        arr = np.asarray(img)
    if arr.shape != dst.shape:
        raise ValueError(f"{path} page {page_index}: decoded {arr.shape}, expected {dst.shape}")
    np.copyto(dst, arr.astype(dst.dtype, copy=False), casting="unsafe")


def select_page(tif: TiffFile, fov_to_z_level: Optional[Dict[int, int]] = None, channel: int = 0) -> int:
    """Which page of the file ``load_single_image`` would return (tile_registration.py:2184-2279)."""
    name = os.path.basename(tif.path)
    n = len(tif.pages)
    if is_ome_tiff_name(name):
        dims = tif.ome_dims()
        if dims is not None:
            z = dims["Z"] // 2 if dims["Z"] > 1 else 0
            if fov_to_z_level:
                info = parse_filename_fov_info(name)
                if info and info["fov"] in fov_to_z_level:
                    z = int(fov_to_z_level[info["fov"]])
            return tif.ome_page(channel=channel, z=z)
        return n // 2 if n > 1 else 0        # the reference's fallback when the OME loader fails (:2236-2247)
    if "_stack" in name:
        return n // 2 if n > 1 else 0
    return 0


def read_tile(path: str, fov_to_z_level: Optional[Dict[int, int]] = None, channel: int = 0) -> np.ndarray:
    """The image ``load_single_image`` returns for this file, without the flatfield step."""
    tif = TiffFile(str(path))
    return tif.read_page(select_page(tif, fov_to_z_level, channel))


def fill_from_tiffs(paths: Sequence[str], fov_to_z_level: Optional[Dict[int, int]] = None, channel: int = 0,
                    workers: int = 8):
    """An ingest ``Fill`` callback (``fill(tile0, tile1, dst)``) that decodes ``paths[tile0:tile1]`` into the
    rows of ``dst``, a view of pinned memory: the strips are read in place, ``workers`` files at a time
    (threads: the work is read() calls, which release the interpreter lock)."""
    paths = [str(p) for p in paths]

    def one(job):
        path, dst = job
        tif = TiffFile(path)
        tif.read_page_into(select_page(tif, fov_to_z_level, channel), dst)

    def fill(lo: int, hi: int, dst: np.ndarray) -> None:
        jobs = [(paths[i], dst[i - lo]) for i in range(lo, hi)]
        if workers <= 1 or len(jobs) <= 1:
            for j in jobs:
                one(j)
            return
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(workers, len(jobs))) as pool:
            list(pool.map(one, jobs))
    return fill
