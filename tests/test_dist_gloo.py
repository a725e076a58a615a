"""The N>1 partitioning and the pair-result all-gather, on CPU with gloo (world_size 2).

The per-rank compute is stood in for by the oracle: what is under test is the
sharding, the halo bookkeeping, the collective and the replicated position solve
between registration and fusion, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from image_stitcher_amd import dist as mdist
from image_stitcher_amd import solve, synth
from oracle import fusion as ofu
from oracle import registration as oreg


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synth.make_grid(5, 3, 3, 128, 128, overlap=0.15, jitter=2)
        rs = mdist.RegistrationShard(g.pair_idx, g.rows, g.cols, world, rank, by="grid_rows")
        # a rank only touches the tiles of its band plus the halo row above it
        local_ids = np.asarray(rs.tile_ids)
        stack = g.tiles[local_ids]
        out, st = oreg.register_pairs(stack, rs.local_pairs(local_ids), g.pair_dir[rs.pair_ids], 32, polish=False)
        g_rows, g_st = rs.gather(torch.from_numpy(out), torch.from_numpy(st))
        # every rank holds every pair now and solves the (tiny) position problem for itself
        row, col = np.repeat(np.arange(g.rows), g.cols), np.tile(np.arange(g.cols), g.rows)
        res = solve.solve_positions(solve.grid_from_pairs(row, col, g.pair_idx, g.pair_dir, g_rows.numpy(),
                                                          g_st.numpy()), 128, 128)
        yx = np.stack([res.y_pos, res.x_pos], axis=1).astype(np.int32)
        assert np.array_equal(yx, g.true_yx - g.true_yx.min(axis=0))
        # fusion bands: each as a mosaic of its own, tile rows relative to the band
        mh, mw = int(yx[:, 0].max()) + 128, int(yx[:, 1].max()) + 128
        fs = mdist.FusionShard(yx[:, 0], 128, mh, world, rank, align=64)
        r0, r1 = fs.row0, fs.row1
        use = fs.tile_ids
        band = ofu.fuse_overwrite(g.tiles[use], yx[use], np.zeros(len(use), dtype=int), 1, mh, mw)[0, r0:r1]
        assert [tuple(v) for v in fs.local_yx(yx)] == [(int(y) - r0, int(x)) for y, x in yx[use]]
        local_tiles = {int(t): None for t in local_ids}
        q.put((rank, g_rows.numpy(), g_st.numpy(), (r0, r1), band, sorted(local_tiles)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = synth.make_grid(5, 3, 3, 128, 128, overlap=0.15, jitter=2)
    ref, ref_st = oreg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 32, polish=False)
    for rank, rows, st, _, _, _ in got:
        assert np.array_equal(rows, ref) and np.array_equal(st, ref_st)  # every rank holds every pair
    assert np.array_equal(ref[:, 1:].astype(np.int64), g.expected_shift)
    # rank 0 owns grid rows 0-1 (tiles 0..5); rank 1 owns row 2 plus the halo row above it
    assert got[0][5] == [0, 1, 2, 3, 4, 5] and got[1][5] == [3, 4, 5, 6, 7, 8]
    yx = (g.true_yx - g.true_yx.min(axis=0)).astype(np.int32)
    mh, mw = int(yx[:, 0].max()) + 128, int(yx[:, 1].max()) + 128
    full = ofu.fuse_overwrite(g.tiles, yx, np.zeros(9, dtype=int), 1, mh, mw)[0]
    stitched = np.concatenate([b for _, _, _, _, b, _ in got])
    assert got[0][3][1] == got[1][3][0] and got[1][3][1] == mh
    assert np.array_equal(stitched, full)


def test_band_helpers():
    assert [mdist.band_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [mdist.shard_mosaic_rows(37065, 8, r, align=2048) for r in range(8)][-1][1] == 37065
    bands = [mdist.shard_mosaic_rows(37065, 8, r, align=2048) for r in range(8)]
    assert all(a[1] == b[0] for a, b in zip(bands[:-1], bands[1:])) and bands[0][0] == 0
    pair_idx, _ = synth.grid_pairs(20, 20)
    sizes = [len(mdist.shard_pairs_by_grid_rows(pair_idx, 20, 20, 8, r)[0]) for r in range(8)]
    assert sum(sizes) == 760 and max(sizes) - min(sizes) <= 39
    even = [mdist.RegistrationShard(pair_idx, 20, 20, 8, r) for r in range(8)]
    assert [len(s.pair_ids) for s in even] == [95] * 8
    assert np.array_equal(np.concatenate([s.pair_ids for s in even]), np.arange(760))
    for s in even:   # the tiles of a run: its centre tiles, their left neighbours and the row above
        need = np.unique(pair_idx[s.pair_ids].reshape(-1))
        assert np.array_equal(s.tile_ids, need) and len(need) <= 4 * 20
        assert np.array_equal(s.local_pairs(need).shape, (95, 2))


def test_rccl_gather_path_with_a_simulated_collective(monkeypatch):
    """The RCCL branch of RegistrationShard.gather (reused buffers, index tables) cannot run on this machine's
    one GPU: its arithmetic is checked with the collective replaced by its definition (rank-major concatenation
    of every rank's padded rows), for every rank of uneven splits, twice in a row."""
    import torch
    import torch.distributed as tdist
    from image_stitcher_amd import dist as mdist, synth
    rows, cols = 5, 4
    pair_idx, pair_dir = synth.grid_pairs(rows, cols)
    n_pairs = len(pair_idx)
    truth = torch.arange(n_pairs * 3, dtype=torch.float64).reshape(n_pairs, 3)
    truth_st = (torch.arange(n_pairs) % 3 == 0).to(torch.int8)
    for world in (2, 3, 8):
        shards = [mdist.RegistrationShard(pair_idx, rows, cols, world, r) for r in range(world)]
        pmax = max(shards[0].counts)
        local = [(truth[torch.as_tensor(s.pair_ids, dtype=torch.long)], truth_st[torch.as_tensor(s.pair_ids, dtype=torch.long)])
                 for s in shards]

        calls = []

        def fake_all_gather(output, inp, _local=local, _pmax=pmax):
            assert inp.dtype == torch.float64 and tuple(inp.shape) == (_pmax, 4)   # rows + status, one buffer
            calls.append(1)
            parts = []
            for r in range(world):
                pad = torch.full((_pmax, 4), float("nan"), dtype=torch.float64)
                pad[:len(_local[r][0]), :3] = _local[r][0]
                pad[:len(_local[r][1]), 3] = _local[r][1]
                parts.append(pad)
            output.copy_(torch.cat(parts))

        monkeypatch.setattr(tdist, "is_initialized", lambda: True)
        monkeypatch.setattr(tdist, "get_backend", lambda *a, **k: "nccl")
        monkeypatch.setattr(tdist, "all_gather_into_tensor", fake_all_gather)
        for r in (0, world - 1):
            for _ in range(2):
                n_before = len(calls)
                got, got_st = shards[r].gather(*local[r])
                assert len(calls) == n_before + 1, "the exchange is ONE collective"
                assert torch.equal(got, truth) and torch.equal(got_st, truth_st), (world, r)
