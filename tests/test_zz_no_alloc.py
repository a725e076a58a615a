"""The allocation-free fusion entry point against the ordinary one (needs an MI355X).

``mistitch_fuse_static`` keeps its per-block tile lists in an array inside the library's code
object instead of a buffer from hipMalloc.  Two calls in a row through the C ABI, the second on a
smaller input (fewer tiles, a smaller mosaic, another mode), must each equal ``mistitch_fuse`` bit
for bit: the shared array keeps the first call's contents and the second must not see them.
"""
import numpy as np
import pytest

from image_stitcher_amd import synth

pytestmark = pytest.mark.gpu


def _fuse(entry, ctx, tiles, yx, plane, ffs, tff, n_planes, mh, mw, mode, rows=None, out=None):
    import torch
    from image_stitcher_amd import _lib
    from image_stitcher_amd._lib import check, ptr
    t, h, w = (int(v) for v in tiles.shape)
    if out is None:
        out = torch.full((n_planes, mh, mw), 0x5a5a, dtype=torch.int16, device=tiles.device)
    ff_type = _lib.FF_NONE if ffs is None else (_lib.FF_F32 if ffs.dtype == torch.float32 else _lib.FF_F64)
    r0, r1 = (0, mh) if rows is None else rows
    check(entry(ctx.handle, ptr(tiles), t, h, w, h * w, ptr(yx), ptr(plane), ptr(ffs), ff_type, ptr(tff), ptr(out),
                n_planes, mh, mw, r0, r1, mode))
    return out


def test_static_workspace_fusion_equals_the_allocating_one():
    import torch
    from image_stitcher_amd import _lib
    from image_stitcher_amd._lib import lib
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream(dev).cuda_stream)

    def case(seed, rows, cols, tile_h, tile_w, n_planes, mode, with_ff, drop=0):
        g = synth.make_grid(seed, rows, cols, tile_h, tile_w, overlap=0.12, jitter=4)
        yx = (g.true_yx - g.true_yx.min(axis=0)).astype(np.int32)
        n = len(g.tiles) - drop
        tiles = torch.from_numpy(np.concatenate([g.tiles[:n]] * n_planes).view(np.int16)).to(dev).view(torch.uint16)
        d_yx = torch.as_tensor(np.tile(yx[:n], (n_planes, 1)), dtype=torch.int32, device=dev)
        plane = torch.as_tensor(np.repeat(np.arange(n_planes), n), dtype=torch.int32, device=dev)
        mh, mw = int(yx[:, 0].max()) + tile_h, int(yx[:, 1].max()) + tile_w
        ffs = tff = None
        if with_ff:
            ffs = torch.as_tensor(np.stack([synth.make_flatfield(tile_h, tile_w, amplitude=0.3 - 0.1 * p)
                                            for p in range(n_planes)]), device=dev)
            tff = plane
        return tiles, d_yx, plane, ffs, tff, n_planes, mh, mw, mode

    # first call: the larger input; second: fewer, smaller tiles, a smaller mosaic, the other mode
    big = case(11, 4, 5, 300, 420, 2, _lib.FUSE_FEATHER, True)
    small = case(12, 2, 3, 130, 97, 1, _lib.FUSE_OVERWRITE, False, drop=1)
    # The property the entry point exists for: no device memory is taken or given back during the calls.  The
    # library's own hipMalloc would bypass torch's caching allocator and show in the driver's free-memory figure;
    # the outputs are allocated beforehand so that torch does not move it either.
    out_big = torch.full((big[5], big[6], big[7]), 0x5a5a, dtype=torch.int16, device=dev)
    out_small = torch.full((small[5], small[6], small[7]), 0x5a5a, dtype=torch.int16, device=dev)
    warm = torch.full((small[5], small[6], small[7]), 0x5a5a, dtype=torch.int16, device=dev)
    _fuse(lib.mistitch_fuse_static, ctx, *small, out=warm)   # code objects loaded, kernels resident
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(dev)[0]
    got_big = _fuse(lib.mistitch_fuse_static, ctx, *big, out=out_big)
    got_small = _fuse(lib.mistitch_fuse_static, ctx, *small, out=out_small)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] == free_before, "device memory was allocated or freed during mistitch_fuse_static"
    want_big = _fuse(lib.mistitch_fuse, ctx, *big)
    want_small = _fuse(lib.mistitch_fuse, ctx, *small)
    assert torch.equal(got_big, want_big)
    assert torch.equal(got_small, want_small)
    assert int((want_big != 0x5a5a).sum().item()) > 0
    # a row band only: rows outside it keep the caller's fill
    band = _fuse(lib.mistitch_fuse_static, ctx, *big, rows=(100, 611))
    assert torch.equal(band[:, 100:611], want_big[:, 100:611])
    assert bool((band[:, :100] == 0x5a5a).all()) and bool((band[:, 611:] == 0x5a5a).all())
    # beyond the workspace: refused with a clear error, nothing written.  65 x 2000 output blocks need
    # 17.2 MB of tile lists; the mosaic passed is REAL (16.8 GB) so that nothing could go out of bounds
    # even if the call were not refused.
    tiles, d_yx, plane, ffs, tff, n_planes, mh, mw, mode = small
    huge_h, huge_w = 32 * 2000, 2048 * 64 + 1000
    n_blocks = ((huge_w + 14) // 8 + 255) // 256 * (huge_h // 32)
    assert n_blocks * 33 * 4 + 16 > (16 << 20)
    huge = torch.full((1, huge_h, huge_w), 0x5a5a, dtype=torch.int16, device=dev)
    rc = lib.mistitch_fuse_static(ctx.handle, _lib.ptr(tiles), int(tiles.shape[0]), 130, 97, 130 * 97, _lib.ptr(d_yx),
                                  _lib.ptr(plane), None, _lib.FF_NONE, None, _lib.ptr(huge), 1, huge_h, huge_w,
                                  0, huge_h, mode)
    assert rc == _lib.ERR_UNSUPPORTED
    assert b"static" in lib.mistitch_last_error()
    torch.cuda.synchronize()
    assert all(bool((huge[:, r:r + 4000] == 0x5a5a).all()) for r in range(0, huge_h, 4000))   # in slices: no 8 GB temporary
    # the allocating entry point takes the same call
    rc = lib.mistitch_fuse(ctx.handle, _lib.ptr(tiles), int(tiles.shape[0]), 130, 97, 130 * 97, _lib.ptr(d_yx),
                           _lib.ptr(plane), None, _lib.FF_NONE, None, _lib.ptr(huge), 1, huge_h, huge_w, 0, huge_h, mode)
    assert rc == _lib.OK
    torch.cuda.synchronize()
    assert torch.equal(huge[0, :mh, :mw], want_small[0]) and int((huge[0, mh:] != 0).sum().item()) == 0
    del huge
    ctx.close()
