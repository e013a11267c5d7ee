"""CPU, world_size 2, gloo: the N>1 path -- ONE input sharded over the ranks by the library's own reader, every rank running the
product's real host pipeline on its shard (the oracle stands in for the HIP kernels, tests/hostpipe.py), the GAF bytes gathered to
rank 0 and re-assembled per segment; the result must be the single-rank bytes (RCCL on the GPU box, same code)."""
import os
import socket
import subprocess
import sys
import tempfile

import pytest

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


def _place_worker(rank, world, port, q, as_tensor):
    """map_sharded's transport alone: every rank owns `world`-dependent, unequal pieces of several output segments (one rank owns nothing at all, one segment is
    empty on every rank); the pieces must land, point to point, at their places in ONE output of exactly the job's size on rank 0"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    from minigraph_amd.dist import map_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def piece(r, s):
        if r == 1 or s == 2:
            return b""
        return (b"<seg%d rank%d>" % (s, r)) * (1 + 5 * r + 3 * s) + b"\n"

    def mapper(r, w):
        n_seg = 5 if r != w - 1 else 3   # the last rank has a SHORTER segment table (its input ran out earlier)
        parts = [piece(r, s) for s in range(n_seg)]
        return b"".join(parts), [len(x) for x in parts]

    got = map_sharded(mapper, dst=0, as_tensor=as_tensor)
    if rank == 0:
        q.put(bytes(got.numpy().tobytes()) if as_tensor else got)
        want_len = sum(len(piece(r, s)) for r in range(world) for s in range(5 if r != world - 1 else 3))
        assert (got.numel() if as_tensor else len(got)) == want_len   # the destination holds every byte once: the output is exactly the job's size
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,as_tensor", [(4, True), (4, False), (3, True)])
def test_pieces_travel_point_to_point_into_place(world, as_tensor):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_place_worker, args=(r, world, port, q, as_tensor)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    def piece(r, s):
        if r == 1 or s == 2:
            return b""
        return (b"<seg%d rank%d>" % (s, r)) * (1 + 5 * r + 3 * s) + b"\n"
    want = b"".join(piece(r, s) for s in range(5) for r in range(world) if s < (5 if r != world - 1 else 3))
    assert got == want


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 23, 1000):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in parts) - min(e - s for s, e in parts) <= 1


def _shard_worker(rank, world, port, q, graph, reads, occ, lco, batch_bases):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import hostpipe as hp
    import minigraph_amd as mga
    from minigraph_amd.dist import map_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def mapper(r, w):  # what mga_map_files_shard does on a GPU: this rank's shard of the ONE input -> GAF bytes + bytes per segment
        shard = reads + ".shard%d.fa" % r
        seg_n = mga.reads_shard_dump(reads, shard, r, w, batch_bases=batch_bases)
        lines, _ = hp.map_with_oracle_stages(graph, shard, occ, lco, per_read=True) if int(seg_n.sum()) else ([], 0)
        seg_len, pos = [], 0
        for n in seg_n:
            seg_len.append(sum(len(x) for x in lines[pos:pos + int(n)]))
            pos += int(n)
        return b"".join(lines), seg_len

    got = map_sharded(mapper, dst=0)
    if rank == 0:
        q.put(got)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gz,world", [(False, 2), (True, 2), (False, 3)])
def test_one_input_n_ranks_one_gaf(gz, world):
    """plain FASTA: the file is cut by byte range at record starts (one segment); gzip: every rank parses everything and keeps its slice
    of every mini-batch (several segments with -K 40k); three ranks: a world size that divides nothing"""
    import gzip
    import hostpipe as hp
    import minigraph_amd as mga
    import refbind as rb
    if not (rb.have_oracle() and os.path.exists(rb.REF_BIN)):
        pytest.skip("oracle/_ref not built")
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "400000", "-H", "3", "-n", "23", "-l", "6000", "-s", "4"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    want, occ, lco = hp.run_reference(graph, reads)
    single, _ = hp.map_with_oracle_stages(graph, reads, occ, lco)
    assert single == want
    if gz:
        open(reads + ".gz", "wb").write(gzip.compress(open(reads, "rb").read()))
        reads += ".gz"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q, graph, reads, occ, lco, 40000)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == want


def _call_worker(rank, world, port, q, graph, reads, occ, lco, out_path):
    """-x asm, contigs sharded over the ranks: every rank maps its contiguous share of the contigs (the product's host pipeline, RMQ chainer included; the oracle stands in for
    the kernels), the mg_gchains_t travel to rank 0 (mga_gchains_pack -> gather -> mga_gchains_unpack), and rank 0 feeds ALL of them, in input order, to the REFERENCE's own
    mg_call_asm (asm-call.c:21: what --call runs behind ggen_map, ggen.c:127-137)"""
    import ctypes as C
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import hostpipe as hp
    import minigraph_amd as mga
    import refbind as rb
    from minigraph_amd.dist import gather_chains
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names, seqs = hp.read_fa(reads)
    L = mga.load()
    # the library's own cut of the file's contigs over the ranks (mga_ggen_shard_range: by bases) and its own assembly on rank 0 (mga_ggen_assemble) -- the two halves of
    # mga_ggen_map_shard that do not need a GPU; the mapping in between is the product's host pipeline with the oracle standing in for the kernels
    from minigraph_amd.dist import ggen_assemble
    ql = (C.c_int * len(names))(*[len(x) for x in seqs])
    b_, e_ = C.c_int(0), C.c_int(0)
    L.mga_ggen_shard_range(len(names), ql, rank, world, C.byref(b_), C.byref(e_))
    st, en = b_.value, e_.value
    shard = reads + ".call%d.fa" % rank
    with open(shard, "wb") as f:
        for i in range(st, en):
            f.write(b">" + names[i] + b"\n" + seqs[i] + b"\n")
    L.mga_gchains_pack.restype = C.c_int64
    L.mga_gchains_pack.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.mga_free.argtypes = [C.c_void_p]
    L.mg_gchain_free.argtypes = [C.c_void_p]
    buf = C.c_void_p()
    if en > st:
        r = hp.map_with_oracle_stages(graph, shard, occ, lco, cigar=True, preset="asm", return_chains=True)
        nb = L.mga_gchains_pack(r["n"], r["gcs"], C.byref(buf))
        for i in range(r["n"]):
            L.mg_gchain_free(r["gcs"][i])
    else:
        nb = L.mga_gchains_pack(0, None, C.byref(buf))
    assert nb > 0
    data = C.string_at(buf, nb)
    L.mga_free(buf)
    parts = gather_bytes(data, dst=0)
    got = ggen_assemble(parts, len(names)) if rank == 0 else None
    if rank == 0:
        assert len(got) == len(names)
        R = rb.Ref().lib

        class bseq1_t(C.Structure):  # mg_bseq1_t, bseq.h:14-17
            _fields_ = [("l_seq", C.c_int32), ("rid", C.c_int32), ("name", C.c_char_p), ("seq", C.c_char_p), ("qual", C.c_char_p), ("comment", C.c_char_p)]
        R.gfa_read.restype = C.c_void_p
        R.mg_call_asm.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        g = R.gfa_read(graph.encode())
        sq = (bseq1_t * len(names))()
        for i in range(len(names)):
            sq[i].l_seq, sq[i].rid, sq[i].name, sq[i].seq = len(seqs[i]), i, names[i], seqs[i]
        arr = (C.c_void_p * len(got))(*got)
        libc = C.CDLL(None)
        libc.fflush(None)
        fd = os.open(out_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        keep = os.dup(1)
        os.dup2(fd, 1)
        try:
            R.mg_call_asm(g, len(names), sq, arr, 5, 100000)   # mg_ggopt_t defaults (options.c:52-54)
            libc.fflush(None)
        finally:
            os.dup2(keep, 1)
            os.close(fd)
            os.close(keep)
        q.put(len(got))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_asm_contigs_sharded_over_ranks_chains_gathered_for_call(world):
    """BASELINE configs[4] / SURVEY 8e: `-cxasm --call` with the contigs sharded over the ranks and the chains gathered: the BED the reference's mg_call_asm prints from the
    gathered objects is what the reference binary prints for the whole file"""
    import hostpipe as hp
    import minigraph_amd as mga
    import refbind as rb
    if not (rb.have_oracle() and os.path.exists(rb.REF_BIN)):
        pytest.skip("oracle/_ref not built")
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "1500000", "-H", "3", "-n", "5", "-l", "400000", "-e", "0.002", "-s", "9"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    p = subprocess.run([rb.REF_BIN, "-cxasm", "--call", "-t", "4", graph, reads], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    want = p.stdout
    assert want.count(b"\n") > 20   # a BED line per bubble and contig
    _, occ, lco = hp.run_reference(graph, reads, cigar=True, preset="asm")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    out_path = os.path.join(d, "call.bed")
    procs = [ctx.Process(target=_call_worker, args=(r, world, port, q, graph, reads, occ, lco, out_path)) for r in range(world)]
    for pr in procs:
        pr.start()
    n = q.get(timeout=600)
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    assert n == 5
    assert open(out_path, "rb").read() == want
