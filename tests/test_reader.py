"""CPU: the FASTA/FASTQ(.gz) reader of mapfiles.c against a straightforward Python parser with kseq's semantics
(bseq.c:61-98 / kseq.h): multi-line records, comments, CRLF, FASTQ, gzip, lower case and U."""
import ctypes as C
import gzip
import os
import random
import tempfile

import minigraph_amd as mga


def fnv(records):
    h = 0xcbf29ce484222325
    for name, seq in records:
        for ch in name + b"\n" + seq + b"\n":
            h = ((h ^ ch) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def norm(seq):
    return seq.replace(b"u", b"t").replace(b"U", b"T").upper()


def parse(path):
    L = mga.load()
    L.mga_reads_parse.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
    n, nb, h = C.c_int64(), C.c_int64(), C.c_uint64()
    assert L.mga_reads_parse(path.encode(), C.byref(n), C.byref(nb), C.byref(h)) == 0
    return n.value, nb.value, h.value


def make(rng, n, fastq, crlf, width):
    nl = b"\r\n" if crlf else b"\n"
    recs, text = [], []
    for i in range(n):
        name = b"read%d" % i
        seq = bytes(rng.choice(b"ACGTacgtNnUu") for _ in range(rng.choice([0, 1, 7, 60, 61, 500, 5000])))
        recs.append((name, norm(seq)))
        text.append((b"@" if fastq else b">") + name + (b" some comment" if i % 3 == 0 else b"") + nl)
        lines = [seq[k:k + width] for k in range(0, len(seq), width)] or [b""]
        text.append(nl.join(lines) + nl)
        if fastq:
            text.append(b"+" + (name if i % 2 else b"") + nl)
            qual = bytes(rng.choice(b"!#>@+5I") for _ in range(len(seq)))  # quality lines may start with '@', '+' or '>'
            text.append(nl.join([qual[k:k + width] for k in range(0, len(qual), width)] or [b""]) + nl)
        if i % 5 == 0 and not fastq:
            text.append(nl)  # stray empty line
    return recs, b"".join(text)


def test_reader_matches_kseq_semantics():
    rng = random.Random(5)
    d = tempfile.mkdtemp()
    for fastq in (False, True):
        for crlf in (False, True):
            for width in (60, 10 ** 9):
                if fastq and width == 60:
                    continue  # multi-line FASTQ is ambiguous by design (kseq handles it by length; covered by width=inf)
                recs, text = make(rng, 200, fastq, crlf, width)
                for gz in (False, True):
                    path = os.path.join(d, "r%d%d%d%d.txt" % (fastq, crlf, width == 60, gz))
                    (gzip.open if gz else open)(path, "wb").write(text)
                    n, nb, h = parse(path)
                    assert (n, nb) == (len(recs), sum(len(s) for _, s in recs)), (fastq, crlf, width, gz)
                    assert h == fnv(recs), (fastq, crlf, width, gz)


def test_reader_large_records_cross_buffer_boundaries():
    rng = random.Random(9)
    d = tempfile.mkdtemp()
    recs = [(b"big%d" % i, bytes(rng.choice(b"ACGT") for _ in range(3_000_000 + i))) for i in range(3)]  # > the 4 MB read buffer in total
    path = os.path.join(d, "big.fa")
    with open(path, "wb") as f:
        for name, seq in recs:
            f.write(b">" + name + b"\n" + seq + b"\n")
    n, nb, h = parse(path)
    assert n == 3 and nb == sum(len(s) for _, s in recs) and h == fnv(recs)


def parse_x(path, batch_bases, threads):
    L = mga.load()
    L.mga_reads_parse_x.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
    n, nb, h = C.c_int64(), C.c_int64(), C.c_uint64()
    assert L.mga_reads_parse_x(path.encode(), batch_bases, threads, C.byref(n), C.byref(nb), C.byref(h)) == 0
    return n.value, nb.value, h.value


def test_parallel_fasta_reader_windows_batches_and_fallback(monkeypatch):
    """the memory-mapped parallel reader (mapfiles.c: fa_*): many windows and mini-batches, several threads, records of very different
    sizes, CRLF, empty records; the same file through the sequential reader (MGA_NO_FAST_READER=1) and a Python parser; and a file
    that turns FASTQ-like half-way (a line starting with '+'), where the rest goes through the sequential reader"""
    rng = random.Random(17)
    d = tempfile.mkdtemp()
    for crlf in (False, True):
        nl = b"\r\n" if crlf else b"\n"
        recs, text = [], []
        for i in range(3000):
            ln = rng.choice([0, 1, 79, 80, 81, 1000, 10000, 10000, 10000, 200000 if i % 500 == 0 else 3000])
            seq = bytes(rng.choices(b"ACGTacgtNu", k=ln))
            name = b"r%d" % i
            recs.append((name, norm(seq)))
            text.append(b">" + name + (b"\tcomment here" if i % 4 == 0 else b"") + nl)
            text.append(b"".join(seq[k:k + 80] + nl for k in range(0, ln, 80)))
        path = os.path.join(d, "p%d.fa" % crlf)
        open(path, "wb").write(b"".join(text))
        want = (len(recs), sum(len(s) for _, s in recs), fnv(recs))
        for bb, th in ((10 ** 9, 1), (10 ** 9, 8), (1_000_000, 4), (100_000, 3), (1, 2)):
            assert parse_x(path, bb, th) == want, (crlf, bb, th)
        monkeypatch.setenv("MGA_NO_FAST_READER", "1")
        assert parse_x(path, 1_000_000, 4) == want
        monkeypatch.delenv("MGA_NO_FAST_READER")
    # FASTA, then FASTQ records from the middle on
    recs, text = [], []
    for i in range(400):
        seq = bytes(rng.choices(b"ACGT", k=5000))
        recs.append((b"a%d" % i, seq))
        text.append(b">a%d\n" % i + b"".join(seq[k:k + 60] + b"\n" for k in range(0, 5000, 60)))
    for i in range(300):
        seq = bytes(rng.choices(b"ACGT", k=700))
        recs.append((b"q%d" % i, seq))
        text.append(b"@q%d\n" % i + seq + b"\n+\n" + bytes(rng.choices(b"!#>@+5I", k=700)) + b"\n")
    path = os.path.join(d, "mixed.fx")
    open(path, "wb").write(b"".join(text))
    want = (len(recs), sum(len(s) for _, s in recs), fnv(recs))
    for bb, th in ((10 ** 9, 4), (300_000, 4)):
        assert parse_x(path, bb, th) == want, (bb, th)


def _dump_records(path):
    recs, name = [], None
    for line in open(path, "rb"):
        if line.startswith(b">"):
            name = line[1:].rstrip(b"\n")
        else:
            recs.append((name, line.rstrip(b"\n")))
    return recs


def test_sharded_reader_any_world_size_reassembles_the_input():
    """ONE input cut for `world` ranks by the library's reader (byte ranges at record starts for a memory-mapped FASTA, read-index slices of
    every mini-batch otherwise): for world sizes that do and do not divide anything, tiny and huge records, CRLF, empty records, more ranks
    than records -- the shards put back in (segment, rank) order are the input, record for record"""
    rng = random.Random(23)
    d = tempfile.mkdtemp()
    for case, (n, crlf, gz) in enumerate([(700, False, False), (700, True, False), (3, False, False), (1, False, False), (250, False, True)]):
        nl = b"\r\n" if crlf else b"\n"
        recs, text = [], []
        for i in range(n):
            ln = rng.choice([0, 1, 59, 60, 61, 2000, 2000, 30000 if i % 97 == 0 else 500])
            seq = bytes(rng.choices(b"ACGTacgtNu", k=ln))
            recs.append((b"r%d" % i, norm(seq)))
            text.append(b">r%d" % i + (b" c=%d" % i if i % 3 == 0 else b"") + nl + b"".join(seq[k:k + 60] + nl for k in range(0, ln, 60)))
        path = os.path.join(d, "s%d.fa" % case + (".gz" if gz else ""))
        (gzip.open if gz else open)(path, "wb").write(b"".join(text))
        for world in (1, 2, 3, 5, 8, 13):
            for bb in ((10 ** 9, 150_000) if n > 10 else (10 ** 9,)):
                shards, segs = [], []
                for r in range(world):
                    out = os.path.join(d, "o%d_%d_%d.fa" % (case, world, r))
                    segs.append([int(x) for x in mga.reads_shard_dump(path, out, r, world, batch_bases=bb, n_threads=3)])
                    shards.append(_dump_records(out))
                    assert sum(segs[-1]) == len(shards[-1])
                got, pos = [], [0] * world
                for s in range(max(len(x) for x in segs)):
                    for r in range(world):
                        k = segs[r][s] if s < len(segs[r]) else 0
                        got += shards[r][pos[r]:pos[r] + k]
                        pos[r] += k
                assert got == recs, (case, world, bb)
