"""Multi-GPU plumbing: reads are sharded across ranks with no data-path collective; the only
exchange is the variable-length gather of GAF bytes to the writer rank.  Backend "nccl" is RCCL over
xGMI on the GPU box; the same code runs on "gloo" CPU tensors in the tests."""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """contiguous, order-preserving split of [0, n_items) (concatenating rank outputs restores input order)"""
    base, rem = divmod(n_items, world)
    st = rank * base + min(rank, rem)
    return st, st + base + (1 if rank < rem else 0)


def gather_bytes(payload, dst=0, device="cpu"):
    """gather one bytes object per rank to `dst`; returns the list in rank order on dst, None elsewhere.
    size all_gather + gather of byte tensors padded to the largest payload"""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=device)
    if len(payload):
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    out = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return [bytes(out[r][:sizes[r]].cpu().numpy().tobytes()) for r in range(world)]
