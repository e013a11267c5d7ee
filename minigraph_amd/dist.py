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


def gather_bytes(payload, dst=0, device="cpu", as_tensors=False):
    """gather one byte string per rank to `dst`; returns the list in rank order on dst, None elsewhere.
    payload: bytes, or any C-contiguous uint8 buffer (e.g. the zero-copy numpy view of the library's GAF buffer).
    One all_gather of the sizes + one gather of byte tensors padded to the largest payload; with as_tensors=True the
    result stays on `device` (uint8 tensors, no host round trip)."""
    import numpy as np
    world, rank = dist.get_world_size(), dist.get_rank()
    src = np.frombuffer(payload, dtype=np.uint8) if not isinstance(payload, np.ndarray) else payload.reshape(-1)
    n = torch.tensor([src.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.empty(mx, dtype=torch.uint8, device=device)
    if src.size:
        if not src.flags.writeable:
            src = src.copy()  # torch.from_numpy wants a writable array (bytes objects are not)
        buf[:src.size].copy_(torch.from_numpy(src), non_blocking=False)
    out = [torch.empty(mx, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    if as_tensors:
        return [out[r][:sizes[r]] for r in range(world)]
    return [out[r][:sizes[r]].cpu().numpy().tobytes() for r in range(world)]
