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
