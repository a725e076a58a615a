"""After one sizing call the registration and fusion entry points enqueue kernels only (no
allocation, no synchronisation, no host copies), so a step can be captured into a HIP graph."""
import numpy as np
import pytest

from image_stitcher_amd import synth

pytestmark = pytest.mark.gpu


def test_step_replayed_from_a_graph_equals_the_eager_step():
    import torch
    from image_stitcher_amd import _lib, fusion, registration
    dev = torch.device("cuda", 0)
    g = synth.make_grid(7, 4, 4, 256, 256, overlap=0.125, jitter=3)
    tiles = registration.to_device_u16(g.tiles)
    ff = torch.as_tensor(synth.make_flatfield(256, 256)[None], device=dev)
    s = torch.cuda.Stream(dev)
    ctx = _lib.Context(0, s.cuda_stream)
    n = len(g.pair_idx)
    out = torch.empty((n, 3), dtype=torch.float64, device=dev)
    status = torch.empty((n,), dtype=torch.int8, device=dev)
    yx = (g.true_yx - g.true_yx.min(axis=0)).astype(np.int32)
    mh, mw = int(yx[:, 0].max()) + 256, int(yx[:, 1].max()) + 256
    d_yx = torch.as_tensor(yx, device=dev)
    d_plane = torch.zeros(len(yx), dtype=torch.int32, device=dev)
    d_idx = torch.as_tensor(g.pair_idx, dtype=torch.int32, device=dev)
    d_dir = torch.as_tensor(g.pair_dir, dtype=torch.uint8, device=dev)
    mosaic = torch.empty((1, mh, mw), dtype=torch.int16, device=dev).view(torch.uint16)

    def step():
        registration.register_pairs(tiles, d_idx, d_dir, 32, ctx=ctx, out=out, status=status, subpixel=True,
                                    flatfield=ff[0])
        fusion.fuse_tiles(tiles, d_yx, d_plane, 1, mh, mw, flatfields=ff, tile_ff=d_plane, mode="feather", out=mosaic,
                          ctx=ctx)

    with torch.cuda.stream(s):
        step()      # sizes every workspace
    torch.cuda.synchronize()
    want_out, want_status, want_mosaic = out.clone(), status.clone(), mosaic.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        step()
    for _ in range(2):
        out.zero_()
        status.zero_()
        mosaic.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want_out) and torch.equal(status, want_status) and torch.equal(mosaic, want_mosaic)
    assert np.array_equal(np.round(want_out.cpu().numpy()[:, 1:]).astype(np.int64), g.expected_shift)


def test_registration_captured_right_after_reserve():
    """mistitch_reserve_registration alone (no warm-up call) must leave nothing to allocate."""
    import torch
    from image_stitcher_amd import _lib, registration
    from image_stitcher_amd._lib import check, lib
    dev = torch.device("cuda", 0)
    g = synth.make_grid(8, 3, 5, 256, 192, overlap=0.125, jitter=3)
    tiles = registration.to_device_u16(g.tiles)
    s = torch.cuda.Stream(dev)
    ctx = _lib.Context(0, s.cuda_stream, pair_chunk=6)     # several passes
    n = len(g.pair_idx)
    flags = _lib.FFT_F64 | _lib.REG_SUBPIXEL
    check(lib.mistitch_reserve_registration(ctx.handle, 256, 192, 32, n, flags))
    out = torch.empty((n, 3), dtype=torch.float64, device=dev)
    status = torch.empty((n,), dtype=torch.int8, device=dev)
    d_idx = torch.as_tensor(g.pair_idx, dtype=torch.int32, device=dev)
    d_dir = torch.as_tensor(g.pair_dir, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        registration.register_pairs(tiles, d_idx, d_dir, 32, ctx=ctx, out=out, status=status, subpixel=True)
    graph.replay()
    torch.cuda.synchronize()
    want, want_st = registration.register_pairs(tiles, d_idx, d_dir, 32, subpixel=True)
    assert torch.equal(status, want_st) and torch.equal(out, want)
