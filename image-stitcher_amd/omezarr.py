"""The writer side of fusion: mosaic and pyramid levels laid out per chunk on the device, then
written as an OME-Zarr (Zarr v2 directory store, uncompressed chunks).

``Stitcher.save_region_ome_zarr`` (stitcher.py:465-650) builds the pyramid with dask, creates one
Zarr array per level with chunks ``(1, 1, 1, min(tile_h, 12288), min(tile_w, 12288))``
(parameters.py:270-272, :695-701) and lets ``da.store`` cut every level into those chunks on the
host; then it writes the multiscales and omero attributes (:596-645).  Here the levels come from
``mistitch_pyramid`` (one read of the mosaic), ``mistitch_chunk_layout`` reorders each level
into whole, zero-padded, C-ordered chunks on the device, and the host's part is one D2H copy per
plane of a level and one ``write()`` per chunk file - no per-chunk slicing or padding in Python.

Scope: ``output_compression="none"`` (chunk file = the chunk's bytes).  The reference's default
compressor (Blosc zstd bitshuffle, :536, :549) is a third-party codec that is not importable here;
a caller that has it can compress the chunk buffers this module hands out (``iter_chunks``).
The metadata follows the OME-NGFF 0.4 layout that ome-zarr's ``write_multiscales_metadata``
produces for the reference's arguments; that library is absent too, so the attribute documents are
checked against the specification's required keys, not against its output (parity unpinned).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import check, lib, ptr

CHUNK_SIZE_LIMIT_PX = 12288     # parameters.py:270-272


def reference_chunks(tile_h: int, tile_w: int) -> Tuple[int, int, int, int, int]:
    """Chunk shape of the output arrays (parameters.py:695-701): one input tile, capped."""
    return (1, 1, 1, min(int(tile_h), CHUNK_SIZE_LIMIT_PX), min(int(tile_w), CHUNK_SIZE_LIMIT_PX))


def default_pyramid_levels(width_px: int, height_px: int, max_dimension: int = 1) -> int:
    """Number of levels when none is configured (parameters.py:1036-1041)."""
    import math
    return max(1, math.ceil(np.log2(max(width_px, height_px) / 1024 * max_dimension)))


def _torch():
    import torch
    return torch


def chunk_layout(image, chunk_h: int, chunk_w: int, ctx=None):
    """``image`` [..., H, W] (uint16 or float64, on the device) -> [planes, ncy, ncx, chunk_h, chunk_w]
    with planes = the product of the leading axes: every [cy, cx] block is one whole chunk of the
    array, zero where it sticks out of the image."""
    torch = _torch()
    if image.dtype not in (torch.uint16, torch.float64):
        raise TypeError(f"uint16 or float64 planes expected, got {image.dtype}")
    if image.dim() < 2 or not image.is_cuda:
        raise ValueError("a device tensor with at least two axes is expected")
    img = image.contiguous()
    h, w = int(img.shape[-2]), int(img.shape[-1])
    planes = int(np.prod(img.shape[:-2])) if img.dim() > 2 else 1
    ncy, ncx = -(-h // chunk_h), -(-w // chunk_w)
    out = torch.empty((planes, ncy, ncx, chunk_h, chunk_w), dtype=img.dtype, device=img.device)
    if out.numel() == 0:
        return out
    c = ctx or _lib.default_context()
    check(lib.mistitch_chunk_layout(c.handle, ptr(img), img.element_size(), planes, h, w, int(chunk_h), int(chunk_w),
                                    ptr(out)))
    return out


def bitshuffle(buf, elem_size: int, block_bytes: int = 1 << 18, ctx=None):
    """Blosc's bitshuffle filter of a device buffer (any shape, viewed as bytes): the first stage of the reference's
    default chunk compressor (Blosc zstd level 5 with BITSHUFFLE, stitcher.py:536, :549).  ``block_bytes`` is the
    codec's block size (the filter works block by block); returns a uint8 tensor of the same byte length.  Chunks
    filtered here go to the entropy coder with its own shuffle switched off."""
    torch = _torch()
    if not buf.is_cuda:
        raise ValueError("a device tensor is expected")
    src = buf.contiguous()
    n_bytes = src.numel() * src.element_size()
    out = torch.empty((n_bytes,), dtype=torch.uint8, device=src.device)
    c = ctx or _lib.default_context()
    check(lib.mistitch_bitshuffle(c.handle, ptr(src), ptr(out), n_bytes, int(elem_size), int(block_bytes)))
    return out


def bitshuffle_numpy(block: np.ndarray, elem_size: int, block_bytes: int = 1 << 18) -> np.ndarray:
    """The filter's definition in NumPy (the check of ``bitshuffle``): per block, the leading whole groups of 8
    elements as a transposed bit matrix (bit k of byte b of element 8 g + i -> row 8 b + k, byte g, bit i), the
    remaining bytes copied."""
    raw = np.ascontiguousarray(block).view(np.uint8).reshape(-1)
    out = np.empty_like(raw)
    for off in range(0, raw.size, block_bytes):
        part = raw[off:off + block_bytes]
        n = (part.size // elem_size) & ~7
        if n:
            bits = np.unpackbits(part[:n * elem_size].reshape(n, elem_size), axis=1, bitorder="little")
            out[off:off + n * elem_size] = np.packbits(bits.T, axis=1, bitorder="little").reshape(-1)
        out[off + n * elem_size:off + part.size] = part[n * elem_size:]
    return out


def multiscales_attrs(name: str, n_levels: int, px_um: float, dz_um: float = 1.0) -> List[dict]:
    """The ``multiscales`` attribute: axes and per-level scale transforms as stitcher.py:489-523, :604-617."""
    datasets = []
    for level in range(n_levels):
        s = 2 ** level
        datasets.append({"path": str(level),
                         "coordinateTransformations": [{"type": "scale",
                                                        "scale": [1, 1, float(dz_um), float(px_um * s), float(px_um * s)]}]})
    axes = [{"name": "t", "type": "time", "unit": "second"},
            {"name": "c", "type": "channel"},
            {"name": "z", "type": "space", "unit": "micrometer"},
            {"name": "y", "type": "space", "unit": "micrometer"},
            {"name": "x", "type": "space", "unit": "micrometer"}]
    return [{"version": "0.4", "name": name, "axes": axes, "datasets": datasets}]


def omero_attrs(name: str, channel_names: Sequence[str], channel_colors: Sequence[int], dtype=np.uint16) -> dict:
    """The ``omero`` attribute exactly as stitcher.py:622-645 fills it."""
    top = int(np.iinfo(dtype).max)
    return {"id": 1, "name": name, "version": "0.4",
            "channels": [{"label": n, "color": f"{int(c):06X}",
                          "window": {"start": 0, "end": top, "min": 0, "max": top},
                          "active": True, "coefficient": 1, "family": "linear"}
                         for n, c in zip(channel_names, channel_colors)]}


def _zarray(shape, chunks, dtype: np.dtype) -> dict:
    return {"zarr_format": 2, "shape": [int(v) for v in shape], "chunks": [int(v) for v in chunks],
            "dtype": np.dtype(dtype).newbyteorder("<").str, "compressor": None, "fill_value": 0 if np.dtype(dtype).kind != "f" else 0.0,
            "filters": None, "order": "C", "dimension_separator": "/"}


def iter_chunks(level_chunks, n_c: int, n_z: int) -> Iterator[Tuple[Tuple[int, int, int, int, int], np.ndarray]]:
    """((t, c, z, cy, cx), chunk) for every chunk of one level already on the host:
    ``level_chunks`` [n_c * n_z, ncy, ncx, chunk_h, chunk_w], plane = c * n_z + z."""
    planes, ncy, ncx = level_chunks.shape[:3]
    if planes != n_c * n_z:
        raise ValueError(f"{planes} planes are not {n_c} channels x {n_z} z levels")
    for p in range(planes):
        for cy in range(ncy):
            for cx in range(ncx):
                yield (0, p // n_z, p % n_z, cy, cx), level_chunks[p, cy, cx]


def write_ome_zarr(path: str, mosaic, channel_names: Sequence[str], channel_colors: Sequence[int], px_um: float,
                   tile_h: int, tile_w: int, n_levels: Optional[int] = None, dz_um: float = 1.0, name: str = "region_t0",
                   ctx=None, writers: int = 8, chunked_hw: Optional[Tuple[int, int]] = None) -> Dict[str, object]:
    """Write ``mosaic`` [C, Z, H, W] uint16 (device) as the OME-Zarr the reference saves for one region
    and timepoint (arrays "0" .. "n_levels-1" of shape (1, C, Z, H >> l, W >> l), level 0 uint16 and
    the others float64 block means, uncompressed).  Returns {"levels": [...shapes], "chunks": (...)}.
    ``chunked_hw`` = (H, W): ``mosaic`` was born in chunk order, [C, Z, ncy, ncx, chunk_h, chunk_w] with the
    reference's chunk shape for these tiles (``fusion.fuse_tiles(chunks=reference_chunks(tile_h, tile_w)[3:])``):
    level 0 goes to the files as it is and the pyramid levels are born in chunk order as well: no layout pass."""
    torch = _torch()
    from . import fusion
    chunks = reference_chunks(tile_h, tile_w)
    ch, cw = chunks[3], chunks[4]
    if chunked_hw is not None:
        if mosaic.dim() != 6 or tuple(int(v) for v in mosaic.shape[-2:]) != (ch, cw):
            raise ValueError(f"a chunk-ordered mosaic must be [C, Z, ncy, ncx, {ch}, {cw}]")
        n_c, n_z = int(mosaic.shape[0]), int(mosaic.shape[1])
        h, w = int(chunked_hw[0]), int(chunked_hw[1])
    else:
        if mosaic.dim() != 4:
            raise ValueError("mosaic must be [C, Z, H, W]")
        n_c, n_z, h, w = (int(v) for v in mosaic.shape)
    if len(channel_names) != n_c or len(channel_colors) != n_c:
        raise ValueError("one name and one colour per channel")
    if n_levels is None:
        n_levels = default_pyramid_levels(w, h)
    # [mosaic, float64 means ...] on the device; the levels of a chunk-ordered mosaic are born in chunk order too
    levels = fusion.generate_pyramid(mosaic, n_levels, chunked_hw=chunked_hw,
                                     level_chunks=(ch, cw) if chunked_hw is not None else None)
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, ".zgroup"), "w") as f:
        json.dump({"zarr_format": 2}, f)
    with open(os.path.join(path, ".zattrs"), "w") as f:
        json.dump({"multiscales": multiscales_attrs(name, n_levels, px_um, dz_um),
                   "omero": omero_attrs(name, channel_names, channel_colors)}, f, indent=2)
    shapes = []
    pool = None
    if writers > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=writers)
    try:
        for l, level in enumerate(levels):
            born_chunked = chunked_hw is not None
            lh, lw = (h >> l, w >> l) if born_chunked else (int(level.shape[-2]), int(level.shape[-1]))
            shape = (1, n_c, n_z, lh, lw)
            shapes.append(shape)
            root = os.path.join(path, str(l))
            os.makedirs(root, exist_ok=True)
            np_dtype = np.uint16 if level.dtype == torch.uint16 else np.float64
            with open(os.path.join(root, ".zarray"), "w") as f:
                json.dump(_zarray(shape, chunks, np_dtype), f, indent=2)
            if lh == 0 or lw == 0:
                continue
            # plane by plane: the chunked copy of a plane (not of the whole level) is what sits on the device
            # and in host memory at any time; the files of one plane are written while the next is laid out
            jobs = []
            for c in range(n_c):
                for z in range(n_z):
                    laid = level[c, z].unsqueeze(0) if born_chunked else chunk_layout(level[c, z], ch, cw, ctx=ctx)
                    host = (laid.view(torch.int16) if laid.dtype == torch.uint16 else laid).cpu().numpy().view(np_dtype)
                    for j in jobs:
                        j.result()
                    jobs = []
                    for (_, _, _, cy, cx), chunk in iter_chunks(host, 1, 1):
                        d = os.path.join(root, "0", str(c), str(z), str(cy))
                        os.makedirs(d, exist_ok=True)
                        target = os.path.join(d, str(cx))
                        if pool is None:
                            chunk.tofile(target)
                        else:
                            jobs.append(pool.submit(chunk.tofile, target))
            for j in jobs:
                j.result()
    finally:
        if pool is not None:
            pool.shutdown()
    return {"levels": shapes, "chunks": chunks}


def read_level(path: str, level: int) -> np.ndarray:
    """Read one array of a store this module wrote (or any uncompressed little-endian Zarr v2 array with '/'
    separated chunk keys) back into [T, C, Z, H, W]: the check of the writer, and a plain statement of the
    on-disk format."""
    root = os.path.join(path, str(level))
    with open(os.path.join(root, ".zarray")) as f:
        meta = json.load(f)
    if meta.get("compressor") is not None or meta.get("filters"):
        raise ValueError("compressed arrays are not read here")
    shape, chunks, dtype = meta["shape"], meta["chunks"], np.dtype(meta["dtype"])
    out = np.full(shape, meta.get("fill_value", 0), dtype=dtype)
    sep = meta.get("dimension_separator", ".")
    grid = [-(-s // c) for s, c in zip(shape, chunks)]
    for idx in np.ndindex(*grid):
        key = sep.join(str(i) for i in idx)
        f = os.path.join(root, *key.split("/")) if sep == "/" else os.path.join(root, key)
        if not os.path.exists(f):
            continue
        block = np.fromfile(f, dtype=dtype).reshape(chunks)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]
    return out
