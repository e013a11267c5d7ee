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
        # kseq drops the CR of a CR LF line end only once the record holds more than that one character (kseq.h ks_getuntil2: "str->l > 1"): a record whose
        # bases are an EMPTY line keeps the CR as its only "base" (checked against the reference's own reader in test_malformed_input_...)
        recs.append((name, norm(seq) if (seq or not crlf) else b"\r"))
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


def _ref_records(path):
    """the reference's own reader (bseq.c:61-98 on kseq.h) through oracle/_ref/libmgref.so: names + normalised bases of every record it yields"""
    import refbind as rb

    class bseq1_t(C.Structure):
        _fields_ = [("l_seq", C.c_int), ("rid", C.c_int), ("name", C.c_char_p), ("seq", C.POINTER(C.c_char)), ("qual", C.c_char_p), ("comment", C.c_char_p)]
    R = C.CDLL(rb.REF_SO)
    R.mg_bseq_open.argtypes = [C.c_char_p]
    R.mg_bseq_open.restype = C.c_void_p
    R.mg_bseq_read.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    R.mg_bseq_read.restype = C.POINTER(bseq1_t)
    R.mg_bseq_close.argtypes = [C.c_void_p]
    fp = R.mg_bseq_open(path.encode())
    assert fp
    out = []
    while True:
        n = C.c_int(0)
        a = R.mg_bseq_read(fp, 1 << 30, 0, 0, 0, C.byref(n))
        if n.value == 0:
            break
        for i in range(n.value):
            out.append((a[i].name, norm(C.string_at(a[i].seq, a[i].l_seq))))
    R.mg_bseq_close(fp)
    return out


def test_malformed_input_is_read_like_kseq_reads_it():
    """inputs kseq does not reject but reads in its own way -- a record marker behind garbage on the same line, a '>' in-line after a FASTQ record,
    a FASTQ record whose quality string is shorter / longer than its bases (kseq_read fails there and bseq.c stops reading the file), lone CR LF lines,
    a file that turns from FASTA into FASTQ -- must give the records the reference's own reader gives (oracle/_ref/libmgref.so: mg_bseq_read)"""
    import pytest
    import refbind as rb
    if not rb.have_ref():
        pytest.skip("oracle/_ref/libmgref.so not built")
    d = tempfile.mkdtemp()
    cases = {
        "garbage_before_marker": b"xx yy>r1 c\nACGT\nAC\n>r2\nGG\n",
        "marker_inline_after_fastq": b"@q1\nACGT\n+\nIIII\njunk>r2\nTTTT\n@q3\nAC\n+\nII\n",
        "fastq_quality_too_short": b"@q1\nACGT\n+\nIIII\n@q2\nACGTACGT\n+\nIII\n@q3\nAC\n+\nII\n",
        "fastq_quality_too_long": b"@q1\nACGT\n+\nIIIIII\n@q2\nAC\n+\nII\n",
        "lone_crlf_lines": b">r1\r\n\r\nACGT\r\n\r\n>r2\r\nA\r\n\r\nC\r\n",
        "cr_only_base_line": b">r1\n\r\nACGT\n>r2\nGG\n",
        "fasta_then_fastq": b">r1\nACGT\n>r2\nGGCC\n@q3\nACGTA\n+\nIIIII\n@q4\nAC\n+\nII\n",
        "empty_fastq_record": b"@q1\n\n+\n\n@q2\nACG\n+\nIII\n",
        "no_trailing_newline": b">r1\nACGT\n>r2\nGG",
    }
    for name, text in cases.items():
        path = os.path.join(d, name + ".fx")
        open(path, "wb").write(text)
        want = _ref_records(path)
        n, nb, h = parse(path)
        assert (n, nb) == (len(want), sum(len(s) for _, s in want)), (name, n, nb, want)
        assert h == fnv(want), (name, want)
