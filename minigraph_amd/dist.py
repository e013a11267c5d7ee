"""Multi-GPU plumbing: ONE input is sharded across ranks (one process per GPU, index replicated) with no data-path collective
(SURVEY 8e); the only exchange is the variable-length gather of GAF bytes to the writer rank, which re-assembles ONE output in input
order.  Backend "nccl" is RCCL over xGMI on the GPU box; the same code runs on "gloo" CPU tensors in the tests."""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """contiguous, order-preserving split of [0, n_items) (concatenating rank outputs restores input order)"""
    base, rem = divmod(n_items, world)
    st = rank * base + min(rank, rem)
    return st, st + base + (1 if rank < rem else 0)


def _pin_for_dma(arr, cap=0):
    """page-lock the memory behind a numpy view of the LIBRARY's GAF buffer (handed back for reuse from step to step, so its address is stable) through the
    library's own registry (mga_host_pin): mga_free() and the writer's realloc unregister before the block moves or goes away.  Best effort: a failure leaves
    the copy on the pageable path."""
    try:
        import ctypes
        import minigraph_amd as mga
        if arr.nbytes < (1 << 20):
            return
        L = mga.load()
        L.mga_host_pin.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.mga_host_pin(arr.ctypes.data, max(int(cap), arr.nbytes))
    except Exception:
        pass


def _as_u8(payload):
    import numpy as np
    return np.frombuffer(payload, dtype=np.uint8) if not isinstance(payload, np.ndarray) else payload.reshape(-1)


def _to_device(src, device, pin=0):
    """the rank's payload as ONE uint8 tensor of exactly its size on `device` (a zero-copy view for host tensors where the array is writable)"""
    t = torch.empty(src.size, dtype=torch.uint8, device=device)
    if src.size:
        copied = False
        if not src.flags.writeable:
            src, copied = src.copy(), True   # torch.from_numpy wants a writable array (bytes objects are not)
        if device != "cpu" and pin and not copied:   # (never register a numpy-owned temporary: `pin` carries the LIBRARY buffer's capacity and lifetime)
            _pin_for_dma(src, pin)   # page-lock the library's (reused) output buffer once: the upload then runs at DMA speed instead of through a bounce buffer
        t.copy_(torch.from_numpy(src), non_blocking=False)
    return t


def _exchange(ops):
    """run a list of point-to-point operations (dist.P2POp) to completion: one group on RCCL (all links into the destination at once), plain isend / irecv on gloo"""
    if not ops:
        return
    for w in dist.batch_isend_irecv(ops):
        w.wait()


def gather_bytes(payload, dst=0, device="cpu", as_tensors=False, pin=False, fail_msg=None):
    """gather one byte string per rank to `dst`; returns the list in rank order on dst, None elsewhere.
    payload: bytes, or any C-contiguous uint8 buffer (e.g. the zero-copy numpy view of the library's GAF buffer).
    One all_gather of the sizes, then SIZE-EXACT point-to-point transfers (round 5): every rank sends exactly its bytes, the destination receives every
    part into a tensor of exactly that part's size -- nothing is padded to the largest payload and the destination holds every byte once.  With as_tensors=True the
    result stays on `device` (uint8 tensors, no host round trip)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    src = _as_u8(payload if payload is not None else b"")
    n = torch.tensor([src.size if payload is not None else -1], dtype=torch.int64, device=device)   # payload None = "this rank failed": said in the one collective every rank is in
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    if min(sizes) < 0:   # every rank raises, nobody is left waiting in a point-to-point transfer
        raise RuntimeError("gather_bytes: rank(s) %s have nothing to send%s" % ([r for r in range(world) if sizes[r] < 0], (": " + fail_msg) if (payload is None and fail_msg) else ""))
    if rank != dst:
        if sizes[rank]:
            _exchange([dist.P2POp(dist.isend, _to_device(src, device, pin), dst)])
        return None
    out = [_to_device(src, device, pin) if r == rank else torch.empty(sizes[r], dtype=torch.uint8, device=device) for r in range(world)]
    _exchange([dist.P2POp(dist.irecv, out[r], r) for r in range(world) if r != rank and sizes[r]])
    if as_tensors:
        return out
    return [t.cpu().numpy().tobytes() for t in out]


def gather_chains(gcs, n, dst=0):
    """-x asm with the contigs of ONE query file sharded over the ranks (BASELINE configs[4]): every rank hands in the mg_gchains_t* of its own contigs
    (`gcs`: ctypes array of n pointers, input order; contiguous shards, rank order = input order); rank `dst` gets the list of ALL ranks' objects in input order --
    rebuilt, malloc-owned copies (mga_gchains_unpack), ready for what consumes a file's mappings (mg_call_asm, mg_ggsimple, mg_cov_asm: ggen.c:39-71,100-137) --
    the other ranks get None.  One pack per rank, one variable-length gather (gather_bytes: RCCL on the GPU box, gloo in the CPU tests)."""
    import ctypes as C
    from . import load
    L = load()
    L.mga_gchains_pack.restype = C.c_int64
    L.mga_gchains_pack.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.mga_gchains_unpack.restype = C.POINTER(C.c_void_p)
    L.mga_gchains_unpack.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_int)]
    L.mga_free.argtypes = [C.c_void_p]
    buf = C.c_void_p()
    nb = L.mga_gchains_pack(n, gcs, C.byref(buf))
    if nb < 0:   # (a failure of ONE rank is raised on all of them, inside gather_bytes' size exchange)
        data, msg = None, "mga_gchains_pack failed: %s" % L.mga_last_error().decode()
    else:
        data, msg = C.string_at(buf, nb), None
        L.mga_free(buf)
    parts = gather_bytes(data, dst=dst, fail_msg=msg)
    if parts is None:
        return None
    out = []
    for p in parts:
        k = C.c_int(0)
        arr = L.mga_gchains_unpack(p, len(p), C.byref(k))
        if not arr:
            raise RuntimeError("mga_gchains_unpack failed: %s" % L.mga_last_error().decode())
        out += [arr[i] for i in range(k.value)]
        L.mga_free(arr)
    return out


def ggen_assemble(parts, n_seq):
    """rank `dst`: the `world` packed buffers (rank order) -> the file's n_seq mg_gchains_t* in input order (the library's mga_ggen_assemble: what ggen_map's r->gcs[] holds
    before mg_call_asm runs, ggen.c:39-71)"""
    import ctypes as C
    from . import load
    L = load()
    L.mga_ggen_assemble.restype = C.POINTER(C.c_void_p)
    L.mga_ggen_assemble.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int]
    L.mga_free.argtypes = [C.c_void_p]
    w = len(parts)
    ptrs = (C.c_char_p * w)(*parts)
    lens = (C.c_int64 * w)(*[len(p) for p in parts])
    arr = L.mga_ggen_assemble(w, ptrs, lens, n_seq)
    if not arr:
        raise RuntimeError("mga_ggen_assemble failed: %s" % L.mga_last_error().decode())
    out = [arr[i] for i in range(n_seq)]
    L.mga_free(arr)
    return out


def ggen_map_sharded(graph, qlens, seqs, names, n_threads=8, dst=0, device="cpu"):
    """`ggen_map` (ggen.c:39-71) over the ranks of a node: every rank maps ITS shard of the file's contigs (mga_ggen_map_shard: the cut is by bases, computed by every rank
    from the lengths alone), the packed chains travel to `dst` size-exact and point to point (RCCL on the GPU box), `dst` gets the file's objects in input order."""
    import ctypes as C
    from . import load
    L = load()
    n = len(qlens)
    L.mga_ggen_map_shard.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.mga_free.argtypes = [C.c_void_p]
    buf, nb = C.c_void_p(), C.c_int64(0)
    ql = (C.c_int * n)(*qlens)
    sq = (C.c_char_p * n)(*seqs)
    nm = (C.c_char_p * n)(*names)
    if L.mga_ggen_map_shard(graph.gi, n, ql, sq, nm, C.byref(graph.mo), n_threads, dist.get_rank(), dist.get_world_size(), C.byref(buf), C.byref(nb)) < 0:
        data, msg = None, "mga_ggen_map_shard failed: %s" % L.mga_last_error().decode()
    else:
        data, msg = C.string_at(buf, nb.value), None
        L.mga_free(buf)
    parts = gather_bytes(data, dst=dst, device=device, fail_msg=msg)
    return None if parts is None else ggen_assemble(parts, n)


def assemble_segments(parts, seg_lens):
    """rank-order concatenation PER SEGMENT: parts[r] = the bytes rank r produced (its segments back to back), seg_lens[r][s] = how many
    of them belong to output segment s (a memory-mapped FASTA file is one segment cut by byte range; any other input has one segment per
    mini-batch cut by read index: mga_map_files_shard in include/minigraph_amd.h).  Returns the single-process output."""
    n_seg = max((len(s) for s in seg_lens), default=0)
    pos = [0] * len(parts)
    out = []
    for s in range(n_seg):
        for r, p in enumerate(parts):
            ln = int(seg_lens[r][s]) if s < len(seg_lens[r]) else 0
            out.append(bytes(memoryview(p)[pos[r]:pos[r] + ln]))
            pos[r] += ln
    return b"".join(out)


def map_sharded(mapper, dst=0, device="cpu", as_tensor=False):
    """one input -> world ranks -> one GAF on `dst` (gmap.c:98-141 fanned out over devices).  mapper(rank, world) maps this rank's shard
    and returns (payload, seg_len[, cap]): its GAF bytes (bytes or a uint8 numpy view) and the per-segment byte counts.  One all_gather of the
    segment tables, then every (segment, rank) piece travels point to point STRAIGHT INTO ITS PLACE in one output tensor of exactly the job's size on `dst`
    (round 5: no padding to the largest shard, no second copy by torch.cat -- the destination holds the output once; on RCCL the seven senders use their own
    xGMI links into the destination at the same time).  Returns the assembled output on dst, None elsewhere: bytes, or with as_tensor=True ONE uint8 tensor on `device`.
    A mapper that returns a third value -- the capacity in bytes of the LIBRARY buffer behind its payload view (MappedGaf.cap) -- gets that buffer page-locked
    once through the library's registry (mga_host_pin) before the upload."""
    world, rank = dist.get_world_size(), dist.get_rank()
    res = mapper(rank, world)
    payload, seg_len = res[0], res[1]
    pin = int(res[2]) if len(res) > 2 else 0   # capacity of the LIBRARY buffer behind the payload view (MappedGaf.cap), if it is one
    seg_len = [int(x) for x in seg_len]
    n_seg = torch.tensor([len(seg_len)], dtype=torch.int64, device=device)
    dist.all_reduce(n_seg, op=dist.ReduceOp.MAX)
    src = _as_u8(payload)
    n_tab = max(int(n_seg.item()), 1)
    tab = torch.zeros(n_tab + 1, dtype=torch.int64, device=device)   # the last word carries the payload's size: every rank can check every table
    if seg_len:
        tab[:len(seg_len)] = torch.tensor(seg_len, dtype=torch.int64)
    tab[n_tab] = int(src.size)
    tabs = [torch.zeros_like(tab) for _ in range(world)]
    dist.all_gather(tabs, tab)
    tabs = [t.cpu().tolist() for t in tabs]
    sizes = [int(t.pop()) for t in tabs]
    bad = [r for r in range(world) if sum(tabs[r]) != sizes[r]]
    if bad:   # a COLLECTIVE failure: every rank sees the same tables and raises before any point-to-point operation is posted (a lone raise would leave dst waiting in irecv)
        raise RuntimeError("map_sharded: rank %d's segment table sums to %d bytes, its payload has %d" % (bad[0], sum(tabs[bad[0]]), sizes[bad[0]]))
    if rank != dst:   # this rank's pieces, in segment order (the order the destination posts its receives for this rank in)
        buf = _to_device(src, device, pin)
        ops, pos = [], 0
        for ln in tabs[rank]:
            if ln:
                ops.append(dist.P2POp(dist.isend, buf[pos:pos + ln], dst))
            pos += ln
        _exchange(ops)
        return None
    total = sum(sum(t) for t in tabs)
    out = torch.empty(total, dtype=torch.uint8, device=device)
    ops, pos, at = [], [0] * world, 0
    own = []
    for s in range(len(tabs[0])):   # output order: segment by segment, ranks in order inside a segment
        for r in range(world):
            ln = int(tabs[r][s])
            if ln and r == rank:
                own.append((at, pos[r], ln))
            elif ln:
                ops.append((r, s, dist.P2POp(dist.irecv, out[at:at + ln], r)))
            pos[r] += ln
            at += ln
    ops.sort(key=lambda x: (x[0], x[1]))   # per source rank in ITS sending order (messages of one pair are matched in order)
    if own:   # the destination's own pieces go from its host buffer straight to their places
        if not src.flags.writeable:
            src = src.copy()
        elif device != "cpu" and pin:
            _pin_for_dma(src, pin)
        hs = torch.from_numpy(src)
        for a_, p_, ln in own:
            out[a_:a_ + ln].copy_(hs[p_:p_ + ln], non_blocking=False)
    _exchange([o[2] for o in ops])
    if as_tensor:
        return out
    return out.cpu().numpy().tobytes()
