"""Device-side mirror of the reference's tile fusion interface.

Mirrors ``image_stitcher/stitcher.py`` (place_tile / place_single_channel_tile /
apply_flatfield_correction / compute_mip / the tile loop of stitch_region) and
``apply_flatfield_to_image`` of ``registration/tile_registration.py`` in the
reference.  The host-side scalar arithmetic that feeds the kernels (pixel
offsets, mosaic size, placement order) is reproduced here in float64 exactly
as the reference evaluates it.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import FF_F32, FF_F64, FF_NONE, FUSE_FEATHER, FUSE_OVERWRITE, check, lib, ptr
from .registration import pixel_max_of, to_device_u16


def _torch():
    import torch
    return torch


# ---- host scalars (stitcher.py:392-397, parameters.py:374-382, 1009-1021) ----
def pixel_size_um(sensor_pixel_size_um: float, magnification: float, tube_lens_f_mm: float,
                  tube_lens_mm: float) -> float:
    return sensor_pixel_size_um / (tube_lens_mm / (tube_lens_f_mm / magnification))


def pixel_offset(pos_mm: float, min_mm: float, px_um: float) -> int:
    return int((pos_mm - min_mm) * 1000 / px_um)


def output_dimensions(x_mm: Sequence[float], y_mm: Sequence[float], tile_h: int, tile_w: int,
                      px_um: float) -> Tuple[int, int]:
    """(width_px, height_px)."""
    width_mm = max(x_mm) - min(x_mm) + (tile_w * px_um / 1000)
    height_mm = max(y_mm) - min(y_mm) + (tile_h * px_um / 1000)
    return int(np.ceil(width_mm * 1000 / px_um)), int(np.ceil(height_mm * 1000 / px_um))


def placement_order(filenames: Sequence[str]) -> np.ndarray:
    """Index permutation that puts tiles in the reference's placement order: sorted
    file paths (parameters.py:614-621), so in overlaps the lexicographically last
    file wins."""
    return np.asarray(sorted(range(len(filenames)), key=lambda i: filenames[i]), dtype=np.int64)


def _ff_type(flatfields) -> int:
    torch = _torch()
    if flatfields is None:
        return FF_NONE
    if flatfields.dtype == torch.float32:
        return FF_F32
    if flatfields.dtype == torch.float64:
        return FF_F64
    raise TypeError(f"flatfield must be float32 or float64, got {flatfields.dtype}")


def _to_device_float(arr, device):
    torch = _torch()
    if isinstance(arr, torch.Tensor):
        return arr.to(device).contiguous()
    a = np.ascontiguousarray(arr)
    if a.dtype not in (np.float32, np.float64):
        raise TypeError(f"flatfield must be float32 or float64, got {a.dtype}")
    return torch.from_numpy(a).to(device)


# ---- kernels -----------------------------------------------------------------
def apply_flatfield_to_image(image, flatfield, dtype=None):
    """``(image / flatfield).clip(0, iinfo(dtype).max).astype(dtype)`` on the device
    (tile_registration.py:2142-2181); ``dtype`` defaults to the image's, uint16 or uint8.  NumPy in -> NumPy out;
    tensors in -> tensor out.  ``image`` may be [H, W] or [N, H, W]; the flatfield plane is shared."""
    torch = _torch()
    is_np = not isinstance(image, torch.Tensor)
    img_shape = tuple(image.shape)
    if tuple(img_shape[-2:]) != tuple(flatfield.shape) or len(img_shape) not in (2, 3):
        raise ValueError(f"Flatfield shape {tuple(flatfield.shape)} doesn't match image shape {img_shape}")
    top = pixel_max_of(image) if dtype is None else {np.dtype(np.uint16): 65535, np.dtype(np.uint8): 255}.get(np.dtype(dtype))
    if top is None:
        raise TypeError("only uint16 and uint8 output is supported")
    t = to_device_u16(image)
    ff = _to_device_float(flatfield, t.device)
    out = torch.empty(t.shape, dtype=torch.int16, device=t.device).view(torch.uint16)
    n = 1 if t.dim() == 2 else int(t.shape[0])
    h, w = (int(s) for s in t.shape[-2:])
    ctx = _lib.default_context()
    with ctx.pixel_max(top):
        check(lib.mistitch_apply_flatfield(ctx.handle, ptr(t), ptr(ff), _ff_type(ff), ptr(out), n, h, w))
    if top == 255:
        out = out.view(torch.int16).to(torch.uint8)
    return out.cpu().numpy() if is_np else out


def compute_mip(tiles):
    """Per-pixel maximum over a list of equal-shape uint16 tiles (stitcher.py:102-121)."""
    torch = _torch()
    if tiles is None or len(tiles) == 0:
        raise ValueError("Cannot compute MIP from empty tile list")
    is_np = not isinstance(tiles[0], torch.Tensor)
    if isinstance(tiles, torch.Tensor):
        stack = to_device_u16(tiles)
    elif is_np:
        stack = to_device_u16(np.stack([np.asarray(t) for t in tiles]))
    else:  # list of tensors: stack through int16 views (uint16 has few torch kernels)
        stack = torch.stack([to_device_u16(t).view(torch.int16) for t in tiles]).view(torch.uint16)
    n = int(stack.shape[0])
    npx = int(stack[0].numel())
    out = torch.empty(stack.shape[1:], dtype=torch.int16, device=stack.device).view(torch.uint16)
    ctx = _lib.default_context()
    check(lib.mistitch_mip(ctx.handle, ptr(stack), n, npx, npx, ptr(out)))
    return out.cpu().numpy() if is_np else out


def generate_pyramid(image, num_levels: int, chunked_hw: Optional[Tuple[int, int]] = None,
                     level_chunks: Optional[Tuple[int, int]] = None):
    """[image, level 1, ..., level num_levels-1] as Stitcher.generate_pyramid builds it
    (stitcher.py:652-668): level l is the float64 mean over 2^l x 2^l blocks of the two
    last axes, excess rows / columns trimmed.  ``image`` is uint16 with any leading axes.
    ``chunked_hw`` = (h, w): ``image`` is a mosaic born in chunk order, ``[..., ncy, ncx, chunk_h, chunk_w]`` of an
    h x w mosaic (``fuse_tiles(chunks=...)``); the levels come out as row-major planes all the same.
    ``level_chunks`` = (chunk_h, chunk_w): the LEVELS come out in chunk order too,
    ``[..., ceil(h_l / chunk_h), ceil(w_l / chunk_w), chunk_h, chunk_w]`` with zeros beyond the level (what
    ``omezarr.chunk_layout`` makes of a row-major level); a chunk-ordered ``image`` must have the same chunk shape."""
    import ctypes as C

    torch = _torch()
    is_np = not isinstance(image, torch.Tensor)
    img = to_device_u16(image)
    if level_chunks is not None:
        ch, cw = int(level_chunks[0]), int(level_chunks[1])
        if ch <= 0 or cw <= 0:
            raise ValueError("chunk shape must be positive")
        if chunked_hw is not None:
            if img.dim() < 4 or tuple(int(v) for v in img.shape[-2:]) != (ch, cw):
                raise ValueError(f"a chunk-ordered mosaic for level chunks of {ch} x {cw} must be [..., ncy, ncx, {ch}, {cw}]")
            h, w = int(chunked_hw[0]), int(chunked_hw[1])
            if tuple(int(v) for v in img.shape[-4:-2]) != (-(-h // ch), -(-w // cw)):
                raise ValueError(f"the chunk grid does not hold a {h} x {w} mosaic")
            lead = tuple(int(v) for v in img.shape[:-4])
        else:
            if img.dim() < 2:
                raise ValueError("image must have at least two axes")
            h, w = int(img.shape[-2]), int(img.shape[-1])
            lead = tuple(int(v) for v in img.shape[:-2])
        planes = int(np.prod(lead)) if lead else 1
        levels = [torch.empty(lead + (-(-(h >> l) // ch), -(-(w >> l) // cw), ch, cw), dtype=torch.float64, device=img.device)
                  for l in range(1, int(num_levels))]
        arr = (C.c_void_p * max(int(num_levels), 1))()
        for l, t in enumerate(levels, start=1):
            arr[l] = t.data_ptr() if t.numel() else None
        check(lib.mistitch_pyramid_to_chunks(_lib.default_context().handle, ptr(img.contiguous()), planes, h, w, ch, cw,
                                             1 if chunked_hw is not None else 0, int(num_levels), arr))
        out = [img] + levels
        return [t.cpu().numpy() for t in out] if is_np else out
    if chunked_hw is not None:
        if img.dim() < 4:
            raise ValueError("a chunk-ordered mosaic has the axes [..., ncy, ncx, chunk_h, chunk_w]")
        h, w = int(chunked_hw[0]), int(chunked_hw[1])
        ncy, ncx, ch, cw = (int(v) for v in img.shape[-4:])
        if (ncy, ncx) != (-(-h // ch), -(-w // cw)):
            raise ValueError(f"{ncy} x {ncx} chunks of {ch} x {cw} do not hold a {h} x {w} mosaic")
        lead = tuple(int(v) for v in img.shape[:-4])
        planes = int(np.prod(lead)) if lead else 1
        levels = [torch.empty(lead + (h >> l, w >> l), dtype=torch.float64, device=img.device)
                  for l in range(1, int(num_levels))]
        arr = (C.c_void_p * max(int(num_levels), 1))()
        for l, t in enumerate(levels, start=1):
            arr[l] = t.data_ptr() if t.numel() else None
        check(lib.mistitch_pyramid_chunked(_lib.default_context().handle, ptr(img.contiguous()), planes, h, w, ch, cw,
                                           int(num_levels), arr))
        out = [img] + levels
        return [t.cpu().numpy() for t in out] if is_np else out
    if img.dim() < 2:
        raise ValueError("image must have at least two axes")
    lead = tuple(int(v) for v in img.shape[:-2])
    h, w = int(img.shape[-2]), int(img.shape[-1])
    planes = int(np.prod(lead)) if lead else 1
    levels = [torch.empty(lead + (h >> l, w >> l), dtype=torch.float64, device=img.device)
              for l in range(1, int(num_levels))]
    arr = (C.c_void_p * max(int(num_levels), 1))()
    for l, t in enumerate(levels, start=1):
        arr[l] = t.data_ptr() if t.numel() else None
    ctx = _lib.default_context()
    check(lib.mistitch_pyramid(ctx.handle, ptr(img), planes, h, w, int(num_levels), arr))
    out = [img] + levels
    return [t.cpu().numpy() for t in out] if is_np else out


def fuse_tiles(tiles, tile_yx, tile_plane, n_planes: int, mosaic_h: int, mosaic_w: int,
               flatfields=None, tile_ff=None, mode: str = "overwrite", rows: Optional[Tuple[int, int]] = None,
               out=None, ctx: Optional[_lib.Context] = None, static_workspace: bool = False,
               chunks: Optional[Tuple[int, int]] = None, pixel_max: Optional[int] = None):
    """Fuse tiles (given IN PLACEMENT ORDER) into ``[n_planes, mosaic_h, mosaic_w]`` uint16.

    tiles      [T, H, W] uint16 (CUDA tensor, or NumPy array that is uploaded)
    tile_yx    [T, 2] int32 mosaic positions (y, x)
    tile_plane [T] int32 destination plane (channel * Z + z)
    flatfields [F, H, W] float32/float64 or None; tile_ff [T] int32 plane per tile (-1 none)
    rows       (row0, row1): only this band of every plane is written (multi-GPU sharding)
    static_workspace  call ``mistitch_fuse_static``: no device allocation inside the library
                      (``out`` should then be given too); fails beyond its size limit
    chunks     (chunk_h, chunk_w): the mosaic is born in the chunk order of the array writer
               (``mistitch_fuse_chunked``): the result is ``[n_planes, ncy, ncx, chunk_h, chunk_w]``, every
               ``[cy, cx]`` one whole zero-padded chunk of a Zarr array with chunks ``(1, 1, 1, chunk_h, chunk_w)``
               (parameters.py:695-701), the same bytes as ``omezarr.chunk_layout`` of the plain mosaic;
               chunk_w must be a multiple of 8
    """
    torch = _torch()
    # 8-bit tiles (or pixel_max=255 for tiles the caller has widened): corrected pixels are clipped to 255 and the mosaic
    # comes back as uint8, the tile type as in the reference (stitcher.py:154-181, :260-267)
    top = pixel_max_of(tiles) if pixel_max is None else int(pixel_max)
    if top == 255 and out is not None:
        raise ValueError("`out` cannot be given with 8-bit tiles: the kernels write a 16-bit mosaic that is narrowed afterwards")
    tiles = to_device_u16(tiles)
    if tiles.dim() != 3:
        raise ValueError(f"tiles must be [T, H, W], got {tuple(tiles.shape)}")
    dev = tiles.device
    n_tiles, h, w = (int(s) for s in tiles.shape)
    yx = torch.as_tensor(np.asarray(tile_yx, dtype=np.int32).reshape(-1, 2) if not isinstance(tile_yx, torch.Tensor)
                         else tile_yx).to(device=dev, dtype=torch.int32).contiguous()
    pl = torch.as_tensor(np.asarray(tile_plane, dtype=np.int32).reshape(-1) if not isinstance(tile_plane, torch.Tensor)
                         else tile_plane).to(device=dev, dtype=torch.int32).contiguous()
    if int(yx.shape[0]) != n_tiles or int(pl.shape[0]) != n_tiles:
        raise ValueError("tile_yx / tile_plane do not match the number of tiles")
    ff = tff = None
    if flatfields is not None:
        ff = _to_device_float(flatfields, dev)
        if ff.dim() == 2:
            ff = ff.unsqueeze(0)
        if tuple(ff.shape[-2:]) != (h, w):
            raise ValueError(f"Flatfield shape {tuple(ff.shape[-2:])} doesn't match image shape {(h, w)}")
        if tile_ff is None:
            raise ValueError("tile_ff is required with flatfields")
        tff = torch.as_tensor(np.asarray(tile_ff, dtype=np.int32).reshape(-1) if not isinstance(tile_ff, torch.Tensor)
                              else tile_ff).to(device=dev, dtype=torch.int32).contiguous()
    row0, row1 = (0, mosaic_h) if rows is None else (int(rows[0]), int(rows[1]))
    mode_id = {"overwrite": FUSE_OVERWRITE, "feather": FUSE_FEATHER}[mode]
    ctx = ctx or _lib.default_context()
    if chunks is not None:
        if static_workspace:
            raise ValueError("the chunk-ordered mosaic has no allocation-free entry point")
        ch, cw = int(chunks[0]), int(chunks[1])
        shape = (n_planes, -(-mosaic_h // ch), -(-mosaic_w // cw), ch, cw)
        if out is None:
            alloc = torch.zeros if rows is not None else torch.empty
            out = alloc(shape, dtype=torch.int16, device=dev).view(torch.uint16)
        elif tuple(out.shape) != shape:
            raise ValueError(f"out must be {shape}, got {tuple(out.shape)}")
        with ctx.pixel_max(top):
            check(lib.mistitch_fuse_chunked(ctx.handle, ptr(tiles), n_tiles, h, w, h * w, ptr(yx), ptr(pl), ptr(ff), _ff_type(ff),
                                            ptr(tff), ptr(out), int(n_planes), int(mosaic_h), int(mosaic_w), ch, cw, row0, row1,
                                            mode_id))
        return out.view(torch.int16).to(torch.uint8) if top == 255 else out
    if out is None:
        alloc = torch.zeros if rows is not None else torch.empty
        out = alloc((n_planes, mosaic_h, mosaic_w), dtype=torch.int16, device=dev).view(torch.uint16)
    entry = lib.mistitch_fuse_static if static_workspace else lib.mistitch_fuse
    with ctx.pixel_max(top):
        check(entry(ctx.handle, ptr(tiles), n_tiles, h, w, h * w, ptr(yx), ptr(pl), ptr(ff), _ff_type(ff),
                    ptr(tff), ptr(out), int(n_planes), int(mosaic_h), int(mosaic_w), row0, row1, mode_id))
    return out.view(torch.int16).to(torch.uint8) if top == 255 else out


# ---- the stitch_region tile loop ---------------------------------------------
@dataclass
class TileInfo:
    """What the reference's AcquisitionMetadata carries for one image file (``fov``: the field-of-view
    index of the key (t, region, fov, z, channel); parsed from ``{region}_{fov}_{z}_...`` when not given)."""
    filename: str
    x_mm: float
    y_mm: float
    z_level: int
    channel: str
    fov: Optional[int] = None

    def fov_index(self) -> int:
        if self.fov is not None:
            return int(self.fov)
        import os
        import re
        m = re.match(r"^[^_]+_(\d+)_\d+_", os.path.basename(self.filename))
        if not m:
            raise ValueError(f"cannot tell the field of view of {self.filename!r}: pass TileInfo.fov")
        return int(m.group(1))


def select_z_layers(strategy, num_z: int) -> list:
    """The z levels a strategy keeps, as ``create_z_layer_selector(strategy).select_z_layers(metadata, num_z)``
    answers (z_layer_selection.py:52-136, :158-200): "all" -> every level, "middle" -> ``[num_z // 2]``
    (ValueError for num_z <= 0), an int or a digit string -> that level (ValueError outside
    ``0 .. num_z - 1``); an unknown string raises ValueError, any other type TypeError.  The result is what
    ``stitch_region(selected_z=...)`` takes."""
    if isinstance(strategy, bool) or not isinstance(strategy, (int, str)):
        raise TypeError(f"Invalid type for z-layer selection strategy: {type(strategy)}. Expected int or str.")
    if isinstance(strategy, str) and strategy.isdigit():
        strategy = int(strategy)
    if isinstance(strategy, int):
        if strategy < 0 or strategy >= num_z:
            raise ValueError(f"Z-layer index {strategy} is out of range. Valid range is 0 to {num_z - 1}")
        return [strategy]
    if strategy == "all":
        return list(range(num_z))
    if strategy == "middle":
        if num_z <= 0:
            raise ValueError(f"Invalid number of z-layers: {num_z}")
        return [num_z // 2]
    raise ValueError(f"Unknown z-layer selection strategy: {strategy}. Available strategies: ['all', 'middle'] or a numeric index")


def z_selection_name(strategy) -> str:
    """The name the reference's selector reports (``get_name``, z_layer_selection.py:77-78, :99-100, :131-132)."""
    if isinstance(strategy, str) and strategy.isdigit():
        strategy = int(strategy)
    if isinstance(strategy, int) and not isinstance(strategy, bool):
        return f"specific_layer_{strategy}"
    if strategy in ("all", "middle"):
        return {"all": "all_layers", "middle": "middle_layer"}[strategy]
    raise ValueError(f"Unknown z-layer selection strategy: {strategy}")


def stitch_region(tiles, infos: Sequence[TileInfo], channel_names: Sequence[str], px_um: float,
                  flatfields: Optional[Dict[int, np.ndarray]] = None, apply_flatfield: bool = False,
                  selected_z: Optional[Sequence[int]] = None, mode: str = "overwrite",
                  update_progress: Optional[Callable[[int, int], None]] = None, apply_mip: bool = False):
    """Device replacement of the tile loop of ``Stitcher.stitch_region`` (stitcher.py:313-410):
    returns the 5-D ``(1, C, Z, H, W)`` CUDA tensor of the tiles' type (uint16, or uint8 for 8-bit tiles).

    tiles[i] belongs to infos[i]; ``tiles`` is ``[T, H, W]``, ``[T, 1, H, W]`` (squeezed, stitcher.py:241-251)
    or ``[T, H, W, 3]`` RGB, whose colours go to the planes ``{base}_R/_G/_B`` of ``channel_names`` with
    ``base = channel.split("_")[0]`` (stitcher.py:224-240).  Placement happens in sorted-filename order,
    the plane of a channel is its position in ``channel_names`` (the reference's monochrome_channels) and z
    levels map to their rank among the selected levels.  ``apply_mip``: the tiles of every (fov, channel)
    group are projected by their per-pixel maximum and placed at z = 0 at the group's first tile's position;
    the output then has one z level (stitcher.py:152, 347-385).  ``flatfields`` is keyed by plane index.
    """
    torch = _torch()
    top = pixel_max_of(tiles)
    tiles = to_device_u16(tiles)
    rgb = tiles.dim() == 4 and int(tiles.shape[-1]) == 3
    if tiles.dim() == 4 and not rgb:
        if int(tiles.shape[1]) != 1:
            raise ValueError(f"Unexpected tile shape: {tuple(tiles.shape[1:])}")
        tiles = tiles[:, 0]
    elif tiles.dim() not in (3, 4):
        raise ValueError(f"Unexpected tile shape: {tuple(tiles.shape[1:])}")
    h, w = (int(s) for s in (tiles.shape[1:3] if rgb else tiles.shape[-2:]))
    zs = sorted(set(i.z_level for i in infos)) if selected_z is None else sorted(set(selected_z))
    z_index = {z: k for k, z in enumerate(zs)}
    keep = [i for i in range(len(infos)) if infos[i].z_level in z_index]
    order = [keep[j] for j in placement_order([infos[i].filename for i in keep])]
    x_min = min(i.x_mm for i in infos)
    y_min = min(i.y_mm for i in infos)
    width, height = output_dimensions([i.x_mm for i in infos], [i.y_mm for i in infos], h, w, px_um)
    names = list(channel_names)
    n_c, n_z = len(names), (1 if apply_mip else len(zs))
    total = len(order)
    as_i16 = tiles.view(torch.int16)

    # what gets placed: (source tensor [H, W(, 3)] uint16, info of the position, output z)
    placed_src, placed_info, placed_z, progress = [], [], [], []
    if apply_mip:
        groups: Dict[Tuple[int, str], List[int]] = {}
        for i in order:
            groups.setdefault((infos[i].fov_index(), infos[i].channel), []).append(i)
        done = 0
        ctx = _lib.default_context()
        for members in groups.values():
            idx = torch.as_tensor(np.asarray(members, dtype=np.int64), device=tiles.device)
            stack = as_i16.index_select(0, idx).view(torch.uint16).contiguous()
            npx = int(stack[0].numel())
            out = torch.empty(stack.shape[1:], dtype=torch.int16, device=stack.device).view(torch.uint16)
            check(lib.mistitch_mip(ctx.handle, ptr(stack), len(members), npx, npx, ptr(out)))
            placed_src.append(out)
            placed_info.append(infos[members[0]])
            placed_z.append(0)
            progress.append(done)
            done += len(members)
        sources = torch.stack([p.view(torch.int16) for p in placed_src]) if placed_src else as_i16[:0]
    else:
        idx = torch.as_tensor(np.asarray(order, dtype=np.int64), device=tiles.device)
        sources = as_i16.index_select(0, idx)
        placed_info = [infos[i] for i in order]
        placed_z = [z_index[infos[i].z_level] for i in order]
        progress = list(range(total))

    # planes: one per mono tile, three per RGB tile (R, G, B placed one after the other, as place_tile does)
    yx_rows, planes = [], []
    for info, z in zip(placed_info, placed_z):
        pos = [pixel_offset(info.y_mm, y_min, px_um), pixel_offset(info.x_mm, x_min, px_um)]
        if rgb:
            base = info.channel.split("_")[0]
            for color in ("R", "G", "B"):
                yx_rows.append(pos)
                planes.append(names.index(f"{base}_{color}") * n_z + z)
        else:
            yx_rows.append(pos)
            planes.append(names.index(info.channel) * n_z + z)
    if rgb:
        sources = sources.permute(0, 3, 1, 2).contiguous().view(-1, h, w)   # [T, 3, H, W] -> [3T, H, W]
    yx = np.asarray(yx_rows, dtype=np.int32).reshape(-1, 2)
    plane = np.asarray(planes, dtype=np.int32)
    ffs = tff = None
    if apply_flatfield and flatfields:
        keys = sorted(flatfields.keys())
        ffs = np.stack([np.asarray(flatfields[k]) for k in keys])
        cidx = plane // n_z
        tff = np.array([keys.index(int(c)) if int(c) in flatfields else -1 for c in cidx], dtype=np.int32)
    mosaic = fuse_tiles(sources.view(torch.uint16), yx, plane, n_c * n_z, height, width, flatfields=ffs, tile_ff=tff,
                        mode=mode, pixel_max=top)
    if update_progress is not None:
        for done in progress:
            update_progress(done, total)
    return mosaic.view(1, n_c, n_z, height, width)
