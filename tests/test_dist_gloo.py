"""The multi-rank exchange (raytracer_amd/dist.py) with world_size 2 on the gloo backend, CPU only.
Rendering is replaced by a deterministic per-pixel function so that the test exercises exactly the
sharding logic: static tile interleave + gather, and dynamic batches + a gather of the tiles each rank rendered."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd.dist import TILE, TileSharder, default_store

W, H = 200, 150   # ragged: 4 x 3 tiles with partial right/bottom tiles


def fake_render(frame, tiles, tiles_x):
    """What a rank would add to its accumulator for `tiles` in `frame`: f(pixel, frame), zero elsewhere."""
    img = torch.zeros((H, W, 4), dtype=torch.float32)
    y, x = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    val = (torch.sin(x * 0.37 + y * 0.11 + frame) + 1.5).to(torch.float32)
    tile_of = (y // TILE) * tiles_x + (x // TILE)
    mask = torch.zeros_like(tile_of, dtype=torch.bool)
    for t in tiles:
        mask |= tile_of == t
    for c in range(3):
        img[..., c] = torch.where(mask, val * (c + 1), torch.zeros_like(val))
    return img


def expected(frames):
    sh = TileSharder(W, H)
    acc = torch.zeros((H, W, 4), dtype=torch.float32)
    for f in frames:
        acc += fake_render(f, range(sh.n_tiles), sh.tiles_x)
    return acc


def worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = TileSharder(W, H, rank, world)
        # static: accumulators live on the owners, gathered once
        first, stride, _ = sh.static_share()
        mine = list(range(first, sh.n_tiles, stride))
        acc = torch.zeros((H, W, 4), dtype=torch.float32)
        for f in (1, 2, 3):
            acc += fake_render(f, mine, sh.tiles_x)
        img = sh.gather_static(acc)
        # dynamic: tiles pulled from the shared counter; rank 0 renders into the accumulator, the others into a
        # frame buffer whose tiles (numbers + pixels) travel to rank 0 and are added there
        store = default_store()
        total = torch.zeros((H, W, 4), dtype=torch.float32)
        buf = torch.zeros((H, W, 4), dtype=torch.float32)
        pulled = 0
        for f in (1, 2, 3):
            tiles = []
            for start, count in sh.dynamic_batches(store, f, 2):
                target = total if rank == 0 else buf
                target += fake_render(f, range(start, start + count), sh.tiles_x)
                tiles += list(range(start, start + count))
                pulled += count
            dist.barrier()
            res = sh.exchange_dynamic(tiles, buf, total)
            assert rank == 0 or not buf.any()          # a sender's buffer is empty again
        cnt = torch.tensor([pulled])
        dist.all_reduce(cnt)
        if rank == 0:
            np.save(out + "_static.npy", img.numpy())
            np.save(out + "_dynamic.npy", res.numpy())
            np.save(out + "_pulled.npy", cnt.numpy())
        else:
            assert img is None and res is None
    finally:
        dist.destroy_process_group()


def test_two_rank_static_gather_and_dynamic_exchange(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "img")
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    want = expected((1, 2, 3)).numpy()
    sh = TileSharder(W, H)
    assert np.array_equal(np.load(out + "_static.npy").view(np.uint32), want.view(np.uint32))
    assert np.array_equal(np.load(out + "_dynamic.npy").view(np.uint32), want.view(np.uint32))
    assert np.load(out + "_pulled.npy")[0] == 3 * sh.n_tiles


def test_tile_ownership():
    sh = TileSharder(W, H)
    assert sh.n_tiles == 12 and TileSharder(W, H, 1, 2).owned().tolist() == [1, 3, 5, 7, 9, 11]


def test_static_plan_with_uneven_shares():
    """12 tiles over 5 ranks: shares of 3, 3, 2, 2, 2 in slots of 3; the exchange of gather_static done by hand
    (pack on every rank, place on rank 0) must rebuild the image exactly and leave the padding in the dummy row."""
    world = 5
    img = torch.rand((H, W, 4))
    n_pix = W * H
    sends = []
    for r in range(world):
        sh = TileSharder(W, H, r, world)
        pack, place = sh._plan("cpu")
        assert (place is None) == (r != 0)
        mine = torch.zeros_like(img)            # a rank's accumulator: its own tiles only
        for t in sh.owned().tolist():
            ty, tx = divmod(t, sh.tiles_x)
            mine[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE] = img[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
        sends.append(mine.reshape(n_pix, 4).index_select(0, pack))
    sh0 = TileSharder(W, H, 0, world)
    _, place = sh0._plan("cpu")
    assert sh0.per_rank == 3 and place.numel() == world * 3 * TILE * TILE
    real = place[place < n_pix]
    assert real.numel() == n_pix and torch.equal(real.sort().values, torch.arange(n_pix))   # every pixel exactly once
    out = torch.empty((n_pix + 1, 4))
    out.index_copy_(0, place, torch.cat(sends))
    assert torch.equal(out[:n_pix].reshape(H, W, 4), img)


def one_rank_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = TileSharder(W, H, 0, 1)
        acc = fake_render(1, range(sh.n_tiles), sh.tiles_x)
        img = sh.gather_static(acc.clone(), force_collective=True)      # the collective calls in a world of one
        same = sh.exchange_dynamic(list(range(sh.n_tiles)), None, acc.clone(), force_collective=True)
        np.save(out + "_one.npy", np.stack([acc.numpy(), img.numpy(), same.numpy()]))
    finally:
        dist.destroy_process_group()


def test_collective_path_in_a_world_of_one(tmp_path):
    """force_collective runs pack -> gather -> place (and the count exchange of the dynamic scheme) with a single
    rank: the image must come back unchanged.  The GPU suite does the same over a one-rank RCCL group."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "img")
    mp.spawn(one_rank_worker, args=(1, port, out), nprocs=1, join=True)
    a, b, c = np.load(out + "_one.npy")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(a.view(np.uint32), c.view(np.uint32))
