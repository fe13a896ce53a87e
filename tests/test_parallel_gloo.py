"""The N > 1 path on CPU: world_size-2 gloo processes exercise the shard arithmetic and the
generation-end all-gather of fitness records (the only collective on the path)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from learning_drones_amd.parallel import shard_bounds, shard_counts

REPO = Path(__file__).resolve().parent.parent


def test_shard_bounds_partition():
    for n in (0, 1, 7, 64, 65536, 262144 + 3):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            c = shard_counts(n, world)
            assert sum(c) == n and max(c) - min(c) <= 1
    assert shard_bounds(262144, 8, 3) == (3 * 32768, 4 * 32768)        # BASELINE config 4
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, str(REPO))
    import torch
    import torch.distributed as dist
    from learning_drones_amd.parallel import gather_records, shard_bounds
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, e = shard_bounds(n_total, world, rank)
        idx = torch.arange(b, e, dtype=torch.int32)
        local = torch.stack([idx * 3 - 1000, idx % 7, idx % 5 | ((idx % 2) << 30), idx], dim=1).contiguous()
        full = gather_records(local, n_total, dist.group.WORLD)
        # every rank then runs the same seeded selection on the same fitness vector
        g = torch.Generator().manual_seed(123)
        pick = torch.randint(0, n_total, (16,), generator=g)
        winners = torch.where(full[pick[::2], 0] > full[pick[1::2], 0], pick[::2], pick[1::2])
        q.put((rank, full.numpy().copy(), winners.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [64, 37])          # equal shards / ragged shards
def test_gather_records_world2(n_total):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort(key=lambda t: t[0])
    idx = np.arange(n_total, dtype=np.int32)
    want = np.stack([idx * 3 - 1000, idx % 7, idx % 5 | ((idx % 2) << 30), idx], axis=1)
    for rank, full, winners in out:
        assert np.array_equal(full, want)
    assert np.array_equal(out[0][2], out[1][2])


def _exchange_worker(rank, world, port, q):
    sys.path.insert(0, str(REPO))
    import torch
    import torch.distributed as dist
    from learning_drones_amd.parallel import DistComm, exchange_parent_genomes
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_local, W = 96, 5                                    # 1.5 genome blocks per rank
        gen = torch.Generator().manual_seed(0)
        whole = torch.randn((world * n_local, W, 4), generator=gen)                      # every rank builds the same
        mine = whole[rank * n_local:(rank + 1) * n_local]
        blocks = (n_local + 63) // 64
        pad = torch.zeros((blocks * 64, W, 4))
        pad[:n_local] = mine
        genomes_local = pad.view(blocks, 64, W, 4).permute(0, 2, 1, 3).contiguous()      # the blocked layout
        pg = torch.Generator().manual_seed(100 + rank)
        parents = torch.randint(0, world * n_local, (n_local,), generator=pg, dtype=torch.int32)
        compact, idx = exchange_parent_genomes(genomes_local, n_local, parents, DistComm(dist.group.WORLD))
        rows = compact.permute(0, 2, 1, 3).reshape(-1, W, 4)                              # back to one row per slot
        q.put((rank, torch.equal(rows[idx.long()], whole[parents.long()]), int(torch.unique(parents).numel()),
               tuple(compact.shape)))
    finally:
        dist.destroy_process_group()


def test_parent_exchange_world2():
    """The all-to-all parent exchange of the multi-GPU generation step on CPU tensors: every child gets
    exactly its parent's genome, and only the distinct needed parents are materialised."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, n_unique, shape in out:
        assert same
        assert shape == ((n_unique + 63) // 64, 5, 64, 4)
