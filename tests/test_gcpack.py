"""mga_gchains_pack / mga_gchains_unpack (csrc/gcpack.c): what moves a rank's mg_gchains_t to the rank that runs --call (ggen.c:39-71).  CPU only: the
chains come from the product's host pipeline with the oracle standing in for the kernels."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import pytest

import minigraph_amd as mga
import hostpipe as hp
import refbind as rb

pytestmark = pytest.mark.skipif(not os.path.exists(rb.REF_BIN), reason="oracle/_ref/minigraph not built")


def _lib():
    L = mga.load()
    L.mga_gchains_pack.restype = C.c_int64
    L.mga_gchains_pack.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.mga_gchains_unpack.restype = C.POINTER(C.c_void_p)
    L.mga_gchains_unpack.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_int)]
    L.mg_gchain_free.argtypes = [C.c_void_p]
    L.mga_free.argtypes = [C.c_void_p]
    L.mg_write_gaf.argtypes = [C.POINTER(hp.kstring_t), C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_char_p, C.c_uint64, C.c_void_p]
    L.gfa_read.restype = C.c_void_p
    return L


def _gaf(L, g, gcs, n, names, seqs, flag):
    ks = hp.kstring_t(0, 0, None)
    out = []
    for i in range(n):
        ql = C.c_int32(len(seqs[i]))
        L.mg_write_gaf(C.byref(ks), g, gcs[i], 1, C.byref(ql), names[i], flag, None)
        out.append(C.string_at(ks.s, ks.l) if ks.l else b"")
    return b"".join(out)


def _workload(d, cigar=True):
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "600000", "-H", "3", "-n", "40", "-l", "6000", "-s", "5"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    with open(reads, "ab") as f:   # reads that get no object at all (map-algo.c:359-360: empty) and no chain (random sequence)
        f.write(b">empty\n\n>noise\n" + b"ACGTTGCA" * 40 + b"\n")
    ref, occ, lco = hp.run_reference(graph, reads, cigar=cigar)
    return graph, reads, ref, occ, lco


def test_pack_unpack_round_trip_prints_the_same_gaf():
    L = _lib()
    d = tempfile.mkdtemp()
    graph, reads, ref, occ, lco = _workload(d)
    r = hp.map_with_oracle_stages(graph, reads, occ, lco, cigar=True, return_chains=True)
    g = L.gfa_read(graph.encode())
    want = _gaf(L, g, r["gcs"], r["n"], r["names"], r["seqs"], r["flag"])
    assert want == ref and want.count(b"\n") >= 30
    buf = C.c_void_p()
    nbytes = L.mga_gchains_pack(r["n"], r["gcs"], C.byref(buf))
    assert nbytes > 0
    data = C.string_at(buf, nbytes)
    L.mga_free(buf)
    for i in range(r["n"]):   # the originals are gone before the copies are read: nothing in the buffer may point into them
        L.mg_gchain_free(r["gcs"][i])
    k = C.c_int(0)
    arr = L.mga_gchains_unpack(data, len(data), C.byref(k))
    assert bool(arr) and k.value == r["n"]
    assert sum(1 for i in range(k.value) if not arr[i]) >= 1   # the empty read has no object on either side
    got = _gaf(L, g, arr, k.value, r["names"], r["seqs"], r["flag"])
    assert got == ref
    buf2 = C.c_void_p()   # packing the unpacked objects gives the same bytes: the format has one representation
    n2 = L.mga_gchains_pack(k.value, arr, C.byref(buf2))
    assert n2 == nbytes and C.string_at(buf2, n2) == data
    L.mga_free(buf2)
    for i in range(k.value):
        L.mg_gchain_free(arr[i])
    L.mga_free(arr)
    open(os.path.join(d, "packed.bin"), "wb").write(data)
    # ---- an untrusted buffer: truncations and corrupted counts / sizes are refused (NULL) or give objects that can be released; never a crash.  In a child process:
    # a segmentation fault must fail this test, not end pytest
    code = r"""
import sys, ctypes as C, random
sys.path.insert(0, %r)
import minigraph_amd as mga
L = mga.load()
L.mga_gchains_unpack.restype = C.POINTER(C.c_void_p); L.mga_gchains_unpack.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_int)]
L.mg_gchain_free.argtypes = [C.c_void_p]; L.mga_free.argtypes = [C.c_void_p]
data = open(sys.argv[1], "rb").read()
rng = random.Random(7)
refused = accepted = 0
def attempt(b):
    global refused, accepted
    k = C.c_int(0)
    arr = L.mga_gchains_unpack(b, len(b), C.byref(k))
    if not arr: refused += 1; return
    accepted += 1
    for i in range(k.value): L.mg_gchain_free(arr[i])
    L.mga_free(arr)
for cut in [0, 1, 8, 15, 16, 17, 47, 48] + [rng.randrange(len(data)) for _ in range(150)] + [len(data) - 1, len(data) - 8]:
    attempt(data[:cut])
assert accepted == 0, accepted          # every proper prefix is refused
for _ in range(400):                    # 32-bit words overwritten with hostile values (counts, sizes, offsets alike)
    b = bytearray(data)
    for _ in range(rng.choice([1, 1, 2, 4])):
        at = rng.randrange(0, len(b) - 4) & ~3
        b[at:at + 4] = rng.choice([b"\xff\xff\xff\x7f", b"\xff\xff\xff\xff", b"\x00\x00\x00\x80", b"\x01\x00\x00\x00", b"\x00\x00\x00\x00", rng.randbytes(4)])
    attempt(bytes(b))
attempt(data + b"\0" * 8)               # trailing bytes: refused or accepted, not read past
r0 = refused
for at in range(8, 200, 4):             # the words that ARE structure: the count behind the magic, the first read's header (present, n_gc, n_lc, n_a) and its first chain record
    for v in (b"\xff\xff\xff\x7f", b"\xff\xff\xff\xff", b"\x00\x00\x00\x80", b"\x00\x00\x10\x00"):   # (whose pointer fields travel as byte counts)
        b = bytearray(data); b[at:at + 4] = v
        attempt(bytes(b))
# ---- ADVICE r4: counts that are consistent with the buffer's SIZE but not with each other.  The first read with chains: header at `at`, gc[] behind it (104 bytes each:
# off @ 8, cnt @ 12, n_anchor @ 16), then lc[] (20 bytes each: off @ 0, cnt @ 4)
import struct
at = 16
while True:
    present, n_gc, n_lc, n_a = struct.unpack_from("<4i", data, at)
    if present and n_gc > 0: break
    at += 32   # (reads without an object / without chains carry a header only)
gc0, lc0 = at + 32, at + 32 + n_gc * 104
def refused_with(patch):
    global refused
    b = bytearray(data); patch(b)
    before = refused
    attempt(bytes(b))
    return refused == before + 1
def put(b, off, v): b[off:off + 4] = struct.pack("<i", v)
assert refused_with(lambda b: put(b, gc0 + 8, n_lc))                      # gc[0].off + cnt beyond lc[]
assert refused_with(lambda b: put(b, gc0 + 12, n_lc + 1))                 # gc[0].cnt
assert refused_with(lambda b: put(b, gc0 + 8, -1))
assert refused_with(lambda b: put(b, lc0 + 0, n_a))                       # lc[0].off + cnt beyond a[]
assert refused_with(lambda b: put(b, lc0 + 4, n_a + 1))
assert refused_with(lambda b: put(b, lc0 + 4, -5))
# an object without chains that still claims records (n_gc = 0, n_lc = 3, n_a = 2: the sizes line up, lc and a would stay NULL behind non-zero counts)
hdr = struct.pack("<QQ", struct.unpack_from("<Q", data, 0)[0], 1)
rec = struct.pack("<8i", 1, 0, 3, 2, 0, 0, 0, 0) + b"\0" * (3 * 20 + 4) + b"\0" * (2 * 16)
before = refused; attempt(hdr + rec); assert refused == before + 1
rec = struct.pack("<8i", 1, 0, 0, 0, 7, 0, 0, 0)                           # ... and the legal form of the same: accepted, no arrays
before = accepted; attempt(hdr + rec); assert accepted == before + 1
print("OK", refused, accepted, refused - r0)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code, os.path.join(d, "packed.bin")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0 and p.stdout.startswith(b"OK"), (p.returncode, p.stderr.decode()[-1500:])
    n_refused, n_accepted, n_struct = (int(x) for x in p.stdout.split()[1:4])
    assert n_refused > 160 and n_struct >= 20, (n_refused, n_accepted, n_struct)   # every prefix, and hostile counts / sizes; a flipped anchor or score word is data, not structure


def test_contigs_are_cut_over_ranks_by_bases():
    """mga_ggen_shard_range: what every rank of a sharded `ggen_map` (ggen.c:39-71) computes from the contig lengths alone -- contiguous, order preserving, every contig
    exactly once, balanced in BASES"""
    import random
    L = mga.load()
    rng = random.Random(3)
    for _ in range(300):
        n = rng.choice([0, 1, 2, 5, 24, 120])
        ql = [rng.choice([0, 1, 1000, 50_000_000, 248_000_000]) if rng.random() < 0.3 else rng.randrange(1, 3_000_000) for _ in range(n)]
        arr = (C.c_int * max(n, 1))(*ql)
        for world in (1, 2, 3, 8):
            pos, sizes = 0, []
            for rank in range(world):
                b, e = C.c_int(-1), C.c_int(-1)
                L.mga_ggen_shard_range(n, arr, rank, world, C.byref(b), C.byref(e))
                assert b.value == pos and e.value >= b.value, (ql, world, rank, b.value, e.value)
                pos = e.value
                sizes.append(sum(ql[b.value:e.value]))
            assert pos == n
            if n and sum(ql):   # no rank carries more than its share plus one contig
                assert max(sizes) <= sum(ql) / world + max(ql) + 1
