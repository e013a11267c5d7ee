"""CPU, world_size 2, gloo: the N>1 path of bench.py -- contiguous read sharding and the gather of
GAF bytes to rank 0 (RCCL on the GPU box, same code)."""
import os
import socket

import torch.distributed as dist
import torch.multiprocessing as mp

from minigraph_amd.dist import gather_bytes, shard_range


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lines = [b"r%d\t10000\t%d\n" % (i, i * 7) for i in range(23)]
    st, en = shard_range(len(lines), rank, world)
    mine = b"".join(lines[st:en]) * (1 + 3 * rank)  # unequal payload sizes
    got = gather_bytes(mine, dst=0)
    empty = gather_bytes(b"" if rank == 1 else b"x", dst=0)
    if rank == 0:
        q.put((got, empty))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, empty = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    lines = [b"r%d\t10000\t%d\n" % (i, i * 7) for i in range(23)]
    assert got[0] == b"".join(lines[:12]) and got[1] == b"".join(lines[12:]) * 4
    assert empty == [b"x", b""]


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 23, 1000):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in parts) - min(e - s for s, e in parts) <= 1
