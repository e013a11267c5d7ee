"""End-to-end parity on the GPU: the whole `-cx lr` job through the C ABI, GAF bytes compared with
the unmodified reference (golden files, and the reference binary itself where oracle/_ref travelled)."""
import hashlib
import os
import zlib
import subprocess
import tempfile

import pytest

import minigraph_amd as mga
import refbind as rb

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def need_ref():
    """the reference binary (oracle/_ref/minigraph, built by oracle/Makefile, shipped with the gpurun snapshot) is the checker
    of these tests: its absence on a GPU box is a FAILURE, never a skip -- a suite that skips its parity checks is not green"""
    assert os.path.exists(rb.REF_BIN), ("oracle/_ref/minigraph is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                        "where /root/reference exists; oracle/_ref/ must travel to the GPU box")


def run_ref(args, out):
    need_ref()
    with open(out, "wb") as fo:
        subprocess.check_call([rb.REF_BIN] + args, stdout=fo, stderr=subprocess.DEVNULL)


def first_diff(a, b):
    la, lb = open(a, "rb").read().split(b"\n"), open(b, "rb").read().split(b"\n")
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            fx, fy = x.split(b"\t"), y.split(b"\t")
            for k, (p, q) in enumerate(zip(fx, fy)):
                if p != q:
                    return "line %d field %d:\n ref: %r\n got: %r" % (i, k, p[:300], q[:300])
            return "line %d: field count %d vs %d" % (i, len(fx), len(fy))
    return "length differs: %d vs %d lines" % (len(la), len(lb))


def test_mt_known_answer():
    """SURVEY/BASELINE known answer: md5 of `-cx lr test/MT.gfa test/MT-orangA.fa`"""
    d = tempfile.mkdtemp()
    out = os.path.join(d, "mt.gaf")
    mga.map_files(os.path.join(GOLD, "MT.gfa"), [os.path.join(GOLD, "MT-orangA.fa")], out)
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == "22bf23ebe2039e8353f56f4a324a2eaa", open(out, "rb").read()[:400]


@pytest.mark.parametrize("cigar", [True, False])
@pytest.mark.parametrize("target", ["gfa", "lin.fa"])
def test_synthetic_vs_reference_binary(cigar, target):
    need_ref()
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "3000000", "-H", "3", "-n", "400", "-s", "5"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t." + target), os.path.join(d, "t.reads.fa")
    ref_out, got = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf")
    run_ref((["-c"] if cigar else []) + ["-x", "lr", "-t", "4", graph, reads], ref_out)
    mga.map_files(graph, [reads], got, cigar=cigar)
    if open(ref_out, "rb").read() != open(got, "rb").read():
        raise AssertionError(first_diff(ref_out, got))


@pytest.mark.parametrize("cigar", [False, True])
def test_single_segment_of_50_Mbp_vs_reference_binary(cigar):
    """BASELINE configs[1] at its target size: a 50 Mbp linear FASTA is ONE segment of 50 Mbp (gfa-io.c:311-322) -- one k_sketch work list of ~800 pieces for the index build, positions
    beyond 2^25 in every anchor, a 50 MB row of the segment images; 3 000 reads (a tenth of them from alt alleles the linear reference does not have) with and without base alignment"""
    need_ref()
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "50000000", "-c", "1", "-H", "3", "-n", "3000", "-s", "11"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.lin.fa"), os.path.join(d, "t.reads.fa")
    assert sum(1 for l in open(graph) if l.startswith(">")) == 1 and os.path.getsize(graph) > 50000000
    ref_out, got = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf")
    run_ref((["-c"] if cigar else []) + ["-x", "lr", "-t", "8", graph, reads], ref_out)
    mga.map_files(graph, [reads], got, cigar=cigar, n_threads=8)
    if open(ref_out, "rb").read() != open(got, "rb").read():
        raise AssertionError(first_diff(ref_out, got))
    assert os.path.getsize(got) > 3000 * (2000 if cigar else 80)
    import shutil
    shutil.rmtree(d, ignore_errors=True)


def test_file_sink_written_in_parallel_slices(monkeypatch):
    """a job whose GAF goes to a regular FILE writes every mini-batch in four slices by pwrite() at their final offsets (mapfiles.c: write_parallel; large outputs only, unless
    MGA_PWRITE_MIN says otherwise): three mini-batches here, the file must be the reference's bytes and end where the last slice ends"""
    need_ref()
    monkeypatch.setenv("MGA_PWRITE_MIN", "1000")
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "3000000", "-H", "3", "-n", "900", "-s", "15"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    ref_out, got = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf")
    run_ref(["-c", "-x", "lr", "-t", "4", "-K", "3500000", graph, reads], ref_out)
    mga.map_files(graph, [reads], got, cigar=True, map_opt=dict(mini_batch_size=3500000))
    if open(ref_out, "rb").read() != open(got, "rb").read():
        raise AssertionError(first_diff(ref_out, got))


@pytest.mark.parametrize("rlen,nreads,min_dev", [(10000, 1500, 100), (40000, 400, 50)])
def test_long_join_rescue_on_device_matches_host_tree_and_reference(monkeypatch, rlen, nreads, min_dev):
    """the RMQ rescue (map-algo.c:407-417) runs inside k_lchain; the sequential AVL tree on the host (MGA_HOST_RESCUE=1)
    and the reference binary must give the same bytes, and the device path must actually have been taken.  Round 6: 10 kb reads re-chain ~250 anchors -- the form with y,
    priority and marks of every anchor in LDS (<= 336 anchors); 40 kb reads re-chain ~1000 -- the form over global memory"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "5000000", "-H", "3", "-n", str(nreads), "-l", str(rlen), "-s", "9", "-S", "77"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
    R = mga.Reads(reads)
    mga.get_stats(G, reset=True)
    dev = mga.map_reads(G, R, n_threads=8)
    st = mga.get_stats(G, reset=True)
    assert st["n_rescue_dev"] > min_dev, st      # bubbles split ~half of the reads into several chains
    assert st["n_rescue_host"] <= st["n_rescue_dev"] // 10, st
    monkeypatch.setenv("MGA_HOST_RESCUE", "1")
    host = mga.map_reads(G, R, n_threads=8)
    st2 = mga.get_stats(G, reset=True)
    assert st2["n_rescue_dev"] == 0
    assert dev == host
    R.close()
    G.close()
    if need_ref() is None:   # (asserts: a GPU box without the reference binary FAILS these tests)
        ref_out = os.path.join(d, "ref.gaf")
        run_ref(["-c", "-x", "lr", "-t", "4", graph, reads], ref_out)
        assert open(ref_out, "rb").read() == dev


@pytest.mark.parametrize("target", ["gfa", "lin.fa"])
def test_device_text_equals_host_text_and_reference(monkeypatch, target):
    """mga_map_reads lets the device stitch the CIGARs and write cg:Z / ds:Z (k_text.hip, incl. the reverse-strand form);
    MGA_HOST_TEXT=1 keeps mg_gchain_cigar / mg_gchain_gen_ds / mg_write_gaf on the host: same bytes, and the reference's"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "4000000", "-H", "3", "-n", "1200", "-s", "21", "-S", "5"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t." + target), os.path.join(d, "t.reads.fa")
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
    R = mga.Reads(reads)
    dev = mga.map_reads(G, R, n_threads=8)
    monkeypatch.setenv("MGA_DEV_GAF", "0")   # round 6: the whole lines come from the device (k_gaf.hip); 0 = the host formats the columns and copies cg / ds behind them
    dev_hostlines = mga.map_reads(G, R, n_threads=8)
    monkeypatch.delenv("MGA_DEV_GAF")
    monkeypatch.setenv("MGA_HOST_TEXT", "1")
    host = mga.map_reads(G, R, n_threads=8)
    R.close()
    G.close()
    assert dev.count(b"\n") > 1000 and b"\tds:Z:" in dev
    assert sum(1 for l in dev.split(b"\n") if l and l.split(b"\t")[4] == b"-") > 100   # reverse-strand lines are exercised
    assert dev == dev_hostlines
    assert dev == host
    if need_ref() is None:   # (asserts: a GPU box without the reference binary FAILS these tests)
        ref_out = os.path.join(d, "ref.gaf")
        run_ref(["-c", "-x", "lr", "-t", "4", graph, reads], ref_out)
        assert open(ref_out, "rb").read() == dev


def test_error_free_reads_need_no_wfa_problem():
    """reads without errors: every gap is a ready '=' operator (galign.c:98-100), the chunk has no WFA problem at all,
    and the text kernel still has to print the chains"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "2000000", "-H", "2", "-n", "300", "-e", "0", "-s", "3"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=4)
    R = mga.Reads(reads)
    dev = mga.map_reads(G, R, n_threads=4)
    R.close()
    G.close()
    import re
    assert len(re.findall(rb"\tcg:Z:\d+=\tds:Z::\d+\n", dev)) > 250   # whole reads in one '=' run (minus the ends before the first / after the last minimizer)
    if need_ref() is None:   # (asserts: a GPU box without the reference binary FAILS these tests)
        ref_out = os.path.join(d, "ref.gaf")
        run_ref(["-c", "-x", "lr", "-t", "4", graph, reads], ref_out)
        assert open(ref_out, "rb").read() == dev


@pytest.mark.parametrize("tag,simargs,cigar", [
    ("50kb reads", ["-G", "5000000", "-H", "3", "-n", "200", "-l", "50000", "-s", "31"], True),
    ("100kb reads, 15% errors", ["-G", "5000000", "-H", "3", "-n", "60", "-l", "100000", "-e", "0.15", "-s", "32"], True),
    ("1kb reads", ["-G", "3000000", "-H", "3", "-n", "3000", "-l", "1000", "-s", "33"], True),
    ("20% errors", ["-G", "3000000", "-H", "3", "-n", "800", "-e", "0.2", "-s", "34"], True),
    ("5 haplotypes, 3 chromosomes", ["-G", "6000000", "-H", "5", "-c", "3", "-n", "1500", "-s", "35"], True),
    ("chains only", ["-G", "3000000", "-H", "3", "-n", "1500", "-s", "36"], False),
    ("1.5 Mbp reads (long-join rescue left to the host tree, strays batched in k_lchain)", ["-G", "12000000", "-H", "3", "-n", "4", "-l", "1500000", "-e", "0.05", "-s", "37"], True),
    ("5 Mbp reads (ultra-long placement: long-query sketch / seeds on the device, first chaining pass on host threads)", ["-G", "16000000", "-H", "3", "-n", "2", "-l", "5000000", "-e", "0.05", "-s", "38"], True),
    ("300 kb reads next to 10 kb reads (chunks of either kind in one job)", ["-G", "8000000", "-H", "3", "-n", "40", "-l", "300000", "-e", "0.06", "-s", "39", "--mix10k"], True),
])
def test_parity_sweep_vs_reference_binary(tag, simargs, cigar, monkeypatch):
    monkeypatch.setenv("MGA_DEV_GCHAIN", "0" if zlib.crc32(tag.encode()) & 1 else "1")  # both placements of graph chaining over the sweep
    monkeypatch.setenv("MGA_GC_SPLIT", "1" if zlib.crc32(tag.encode()) & 2 else "0")     # ... and, on the device, both its one-kernel and its three-launch form
    """shapes the benchmark workload does not reach: wide WFA tiers (long gaps of long / noisy reads), many short reads,
    several stable sequences, the chains-only output"""
    need_ref()
    d = tempfile.mkdtemp()
    mix = "--mix10k" in simargs
    simargs = [x for x in simargs if x != "--mix10k"]
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t")] + simargs, stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    if mix:   # ordinary reads of the same graph interleaved with the long ones: a job whose chunks alternate between the two placements
        subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "s")] + [("400" if simargs[i - 1] == "-n" else "10000" if simargs[i - 1] == "-l" else x) for i, x in enumerate(simargs)], stderr=subprocess.DEVNULL)
        long_recs = open(reads, "rb").read().split(b">")[1:]
        short_recs = [r.replace(b"r", b"s", 1) for r in open(os.path.join(d, "s.reads.fa"), "rb").read().split(b">")[1:]]
        with open(reads, "wb") as f:
            for i, r in enumerate(short_recs):
                f.write(b">" + r)
                if i % 10 == 9 and i // 10 < len(long_recs):
                    f.write(b">" + long_recs[i // 10])
    ref_out = os.path.join(d, "ref.gaf")
    run_ref((["-c"] if cigar else []) + ["-x", "lr", "-t", "8", graph, reads], ref_out)
    G = mga.Graph(graph, preset="lr", cigar=cigar, n_threads=8)
    R = mga.Reads(reads)
    got = mga.map_reads(G, R, n_threads=8)
    R.close()
    G.close()
    if open(ref_out, "rb").read() != got:
        open(os.path.join(d, "got.gaf"), "wb").write(got)
        raise AssertionError(tag + ": " + first_diff(ref_out, os.path.join(d, "got.gaf")))


def test_reads_at_the_long_read_boundary_vs_reference_binary():
    """reads of exactly MGA_LONG_READ - 2 .. + 1 bases (262 142 .. 262 145): the last two that go through k_lchain and the first two whose first chaining pass runs on host
    threads (hchain.c), in one job and in both orders -- the same bytes as the reference whichever side of the switch a read falls on"""
    need_ref()
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "6000000", "-H", "3", "-n", "8", "-l", "300000", "-e", "0.05", "-s", "91"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    recs = []
    for r in open(reads, "rb").read().split(b">")[1:]:
        name, _, seq = r.partition(b"\n")
        recs.append((name, seq.replace(b"\n", b"")))
    assert len(recs) == 8 and min(len(q) for _, q in recs) >= 262145
    lens = [262143, 262144, 262142, 262145, 262145, 262142, 262144, 262143]
    with open(reads, "wb") as f:
        for (name, q), n in zip(recs, lens):
            f.write(b">" + name + b"\n" + q[:n] + b"\n")
    ref_out = os.path.join(d, "ref.gaf")
    run_ref(["-c", "-x", "lr", "-t", "8", graph, reads], ref_out)
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
    R = mga.Reads(reads)
    got = mga.map_reads(G, R, n_threads=8)
    R.close()
    G.close()
    if open(ref_out, "rb").read() != got:
        open(os.path.join(d, "got.gaf"), "wb").write(got)
        raise AssertionError(first_diff(ref_out, os.path.join(d, "got.gaf")))


def test_edge_case_reads_vs_reference_binary():
    """the shapes the reference's own callers have to survive: empty and tiny reads, reads without a single
    minimizer hit, runs of N, lower case, U, FASTQ input, duplicated names"""
    need_ref()
    import random
    rng = random.Random(17)
    human = b"".join(l.strip() for l in open(os.path.join(GOLD, "MT-human.fa"), "rb") if not l.startswith(b">")).upper()
    comp = bytes.maketrans(b"ACGT", b"TGCA")

    def noisy(s, e):
        out = bytearray()
        for ch in s:
            r = rng.random()
            if r < e * 0.4:
                out.append(rng.choice(b"ACGT"))
            elif r < e * 0.7:
                out.append(ch); out.append(rng.choice(b"ACGT"))
            elif r < e:
                pass
            else:
                out.append(ch)
        return bytes(out)

    recs = [
        (b"empty", b""),
        (b"tiny", human[100:110]),
        (b"below_k", human[200:216]),
        (b"allN", b"N" * 500),
        (b"random", bytes(rng.choice(b"ACGT") for _ in range(5000))),
        (b"clean", human[1000:6000]),
        (b"noisy", noisy(human[2000:9000], 0.12)),
        (b"revcomp", noisy(human[3000:8000], 0.1).translate(comp)[::-1]),
        (b"withN", noisy(human[500:3000], 0.05) + b"N" * 40 + noisy(human[3040:7000], 0.05)),
        (b"lower", noisy(human[4000:9000], 0.08).lower()),
        (b"rna", noisy(human[6000:9000], 0.05).replace(b"T", b"U")),
        (b"clean", human[7000:12000]),                       # duplicated name
        (b"chimera", noisy(human[1000:4000], 0.05) + noisy(human[9000:12000], 0.05).translate(comp)[::-1]),
        (b"long", noisy(human, 0.1)),
    ]
    d = tempfile.mkdtemp()
    fa, fq = os.path.join(d, "e.fa"), os.path.join(d, "e.fq")
    with open(fa, "wb") as f:
        for n, s in recs:
            f.write(b">" + n + b" comment\n" + b"\n".join(s[k:k + 70] for k in range(0, len(s), 70)) + b"\n")
    with open(fq, "wb") as f:
        for n, s in recs:
            f.write(b"@" + n + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
    graph = os.path.join(GOLD, "MT.gfa")
    for reads in (fa, fq):
        for cigar in (True, False):
            ref_out, got = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf")
            run_ref((["-c"] if cigar else []) + ["-x", "lr", "-t", "2", graph, reads], ref_out)
            mga.map_files(graph, [reads], got, cigar=cigar)
            if open(ref_out, "rb").read() != open(got, "rb").read():
                raise AssertionError("%s cigar=%s: %s" % (os.path.basename(reads), cigar, first_diff(ref_out, got)))


def test_device_index_build_equals_host_build(monkeypatch):
    """mg_index builds the minimizer table on the device (k_index.hip: sketch -> stable radix sort -> CAS insertion);
    MGA_HOST_INDEX=1 keeps the sequential host build: same occurrence thresholds, same mapping bytes"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "8000000", "-H", "4", "-n", "600", "-s", "41"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    out = {}
    for host in ("0", "1"):
        monkeypatch.setenv("MGA_HOST_INDEX", host)
        G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
        R = mga.Reads(reads)
        out[host] = (G.mo.occ_max1, G.mo.lc_max_occ, mga.map_reads(G, R, n_threads=8))
        R.close()
        G.close()
    assert out["0"][:2] == out["1"][:2]
    assert out["0"][2] == out["1"][2] and out["0"][2].count(b"\n") >= 600
    # anchored to the reference: mg_index + mg_opt_update of libmgref.so on the same graph give the same occurrence thresholds
    # (mg_idx_cal_quantile, index.c:74-93 over ITS hash tables), and the mapping bytes are the reference binary's
    import ctypes as C
    need_ref()
    L = rb.Ref().lib
    L.mg_opt_set.argtypes = [C.c_char_p, C.POINTER(mga.idxopt_t), C.POINTER(mga.mapopt_t), C.POINTER(mga.ggopt_t)]
    L.mg_index.argtypes = [C.c_void_p, C.POINTER(mga.idxopt_t), C.c_int, C.POINTER(mga.mapopt_t)]
    L.mg_index.restype = C.c_void_p
    L.mg_idx_destroy.argtypes = [C.c_void_p]
    io, mo, go = mga.idxopt_t(), mga.mapopt_t(), mga.ggopt_t()
    L.mg_opt_set(None, C.byref(io), C.byref(mo), C.byref(go))
    L.mg_opt_set(b"lr", C.byref(io), C.byref(mo), C.byref(go))
    g = L.gfa_read(graph.encode())
    gi = L.mg_index(g, C.byref(io), 4, C.byref(mo))
    assert (mo.occ_max1, mo.lc_max_occ) == out["0"][:2], ((mo.occ_max1, mo.lc_max_occ), out["0"][:2])
    L.mg_idx_destroy(gi)
    L.gfa_destroy(g)
    ref_out = os.path.join(d, "ref.gaf")
    run_ref(["-c", "-x", "lr", "-t", "8", graph, reads], ref_out)
    assert open(ref_out, "rb").read() == out["0"][2]


@pytest.mark.parametrize("preset", ["lr", "asm"])
@pytest.mark.parametrize("query", ["MT-human.fa", "MT-chimp.fa", "MT-orangA.fa"])
def test_reference_fixtures_both_presets(preset, query):
    """the reference's own fixtures (test/MT*.fa vs test/MT.gfa) under -x lr and -x asm (RMQ chainer as the primary chainer,
    lchain.c:252-372), with and without base alignment"""
    need_ref()
    d = tempfile.mkdtemp()
    for cigar in (True, False):
        ref_out, got = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf")
        run_ref((["-c"] if cigar else []) + ["-x", preset, "-t", "2", os.path.join(GOLD, "MT.gfa"), os.path.join(GOLD, query)], ref_out)
        mga.map_files(os.path.join(GOLD, "MT.gfa"), [os.path.join(GOLD, query)], got, preset=preset, cigar=cigar)
        if open(ref_out, "rb").read() != open(got, "rb").read():
            raise AssertionError("%s %s cigar=%s: %s" % (preset, query, cigar, first_diff(ref_out, got)))


@pytest.mark.parametrize("contig,n,err", [(300000, 30, 0.005), (4000000, 5, 0.001)])
def test_asm_preset_long_contigs_vs_reference_binary(contig, n, err, monkeypatch):
    """-cx asm on assembly-like queries: 300 kb and 4 Mbp contigs against the bubble graph.  This is the long-query path: sketch in
    64 kb pieces, one thread per minimizer for the seeds, anchors sorted by the host, the RMQ chainer's forward pass spread over
    (segment, strand) runs.  MGA_NO_LONGQ=1 (one wavefront per contig, device sort) must give the same bytes"""
    need_ref()
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "6000000", "-H", "3", "-n", str(n), "-l", str(contig), "-e", str(err), "-s", "51"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    ref_out, got, got2 = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf"), os.path.join(d, "got2.gaf")
    run_ref(["-c", "-x", "asm", "-t", "8", graph, reads], ref_out)
    mga.map_files(graph, [reads], got, preset="asm", cigar=True)
    if open(ref_out, "rb").read() != open(got, "rb").read():
        raise AssertionError(first_diff(ref_out, got))
    if contig <= 300000:
        monkeypatch.setenv("MGA_NO_LONGQ", "1")
        mga.map_files(graph, [reads], got2, preset="asm", cigar=True)
        assert open(got2, "rb").read() == open(got, "rb").read()


@pytest.mark.parametrize("tag,simargs", [
    ("4 Mbp contigs, 0.1 % divergence", ["-G", "12000000", "-H", "3", "-n", "4", "-l", "4000000", "-e", "0.001", "-s", "61"]),
    ("1 Mbp contigs, 3 % divergence (inexact predecessors: the inner window's ordered walk)", ["-G", "8000000", "-H", "4", "-n", "6", "-l", "1000000", "-e", "0.03", "-s", "62"]),
    ("300 kb contigs, 8 % divergence, 5 haplotypes", ["-G", "6000000", "-H", "5", "-n", "12", "-l", "300000", "-e", "0.08", "-s", "63"]),
])
def test_rmq_forward_pass_on_device_equals_host_tree_and_reference(tag, simargs, monkeypatch):
    """round 5 (VERDICT r4 missing 1): the RMQ chainer's forward pass (mg_lchain_rmq, lchain.c:252-357), the primary chainer of -x asm, runs on the device -- a
    wavefront per (segment, strand) run, the tree's range-minimum query as an arg-min with tie DETECTION, the inner window as a rank-sorted replay (k_rmq.hip) -- and gives
    the bytes of the host's exact AVL tree (MGA_DEV_RMQ=0) and of the reference; the counters say the device really took the runs and how many it handed back"""
    import ctypes as C
    need_ref()
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t")] + simargs, stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    ref_out, got_dev, got_host = os.path.join(d, "ref.gaf"), os.path.join(d, "dev.gaf"), os.path.join(d, "host.gaf")
    run_ref(["-c", "-x", "asm", "-t", "8", graph, reads], ref_out)
    L = mga.load()
    st = (C.c_int64 * 8)()
    L.mga_rq_dev_stats(st, 1)
    mga.map_files(graph, [reads], got_dev, preset="asm", cigar=True)
    L.mga_rq_dev_stats(st, 1)
    n_dev, n_tie, n_big, n_long, n_fail = st[0], st[1], st[2], st[3], st[4]
    if open(ref_out, "rb").read() != open(got_dev, "rb").read():
        raise AssertionError(tag + " (device): " + first_diff(ref_out, got_dev))
    assert n_dev > 0 and n_fail == 0, (n_dev, n_tie, n_big, n_long, n_fail)
    assert n_dev >= 20 * (n_tie + n_big), (n_dev, n_tie, n_big)   # handing runs back is the exception
    monkeypatch.setenv("MGA_DEV_RMQ", "0")
    mga.map_files(graph, [reads], got_host, preset="asm", cigar=True)
    L.mga_rq_dev_stats(st, 1)
    assert st[0] == 0
    assert open(got_host, "rb").read() == open(got_dev, "rb").read()


def test_reference_shaped_c_api_mg_map_and_mg_map_batch():
    """what a caller of minigraph.h does: mg_map_batch() / mg_map() -> mg_gchains_t -> mg_write_gaf() ->
    mg_gchain_free(); CIGAR stitching and ds run on the host on this path.  Same bytes as the GAF-only path and the reference"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "3000000", "-H", "3", "-n", "300", "-s", "61"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    names, seqs = [], []
    for line in open(reads, "rb"):
        if line.startswith(b">"):
            names.append(line[1:].split()[0])
        else:
            seqs.append(line.strip())
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
    R = mga.Reads(reads)
    want = mga.map_reads(G, R, n_threads=8)
    R.close()
    batch = mga.map_batch_api(G, names, seqs, n_threads=8)
    single = mga.map_batch_api(G, names[:40], seqs[:40], per_read=True)
    G.close()
    assert batch == want
    assert single == b"".join(want.split(b"\n")[k] + b"\n" for k in range(40))
    if need_ref() is None:   # (asserts: a GPU box without the reference binary FAILS these tests)
        ref_out = os.path.join(d, "ref.gaf")
        run_ref(["-c", "-x", "lr", "-t", "4", graph, reads], ref_out)
        assert open(ref_out, "rb").read() == batch


def test_pipeline_knobs_do_not_change_the_output(monkeypatch):
    """chunking, pipeline depth, host threads and the tier scheduling mode are performance knobs only"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "3000000", "-H", "3", "-n", "257", "-s", "71"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
    R = mga.Reads(reads)
    want = mga.map_reads(G, R, n_threads=8)
    L = mga.load()
    for chunk, pipe, threads, extra in ((1, 4, 2, {}), (3, 2, 1, {}), (64, 1, 8, {}), (100, 3, 16, {"MGA_WFA_CONCURRENT": "1"}), (7, 4, 3, {"MGA_UPLOAD_READS": "1"}),
                                        (40, 4, 8, {"MGA_CUT": "0"}), (40, 4, 8, {"MGA_CUT": "1", "MGA_TAIL": "3"}), (64, 3, 8, {"MGA_TAIL": "2", "MGA_WFA_GRID_PCT": "50", "MGA_WFA_SLOTS": "3"}),
                                        (50, 2, 4, {"MGA_RAMP": "0", "MGA_WFA_GRID_PCT": "1"}), (40, 4, 8, {"MGA_GAF_DIRECT": "0", "MGA_DEV_SPLICE": "0"}),
                                        (23, 6, 5, {"MGA_GAF_DIRECT": "1", "MGA_DEV_SPLICE": "1", "MGA_FRONT_SLOTS": "2", "MGA_WFA_TB_SIDE": "1"}), (64, 4, 8, {"MGA_DEV_GCHAIN": "1", "MGA_DEV_SPLICE": "0"}),
                                        (33, 8, 3, {"MGA_DEV_GCHAIN": "1", "MGA_GAF_DIRECT": "0"}),
                                        # round 6: a share of the chunks chained on the device, the rest on the host threads (only with > 12 of them)
                                        (9, 6, 16, {"MGA_DEV_GCHAIN_PCT": "50"}), (5, 4, 14, {"MGA_DEV_GCHAIN_PCT": "30", "MGA_DEV_GAF": "0"}), (13, 3, 16, {"MGA_DEV_GCHAIN_PCT": "100"})):
        monkeypatch.setenv("MGA_CHUNK", str(chunk))
        monkeypatch.setenv("MGA_PIPE", str(pipe))
        for k, v in extra.items():
            monkeypatch.setenv(k, v)
        L.mga_idx_stream_close(G.gi)   # the index's chunk pipeline is rebuilt with THESE knobs (chunk size, pipeline threads and the cut are fixed when a stream is opened)
        got = mga.map_reads(G, R, n_threads=threads)
        for k in extra:
            monkeypatch.delenv(k)
        assert got == want, (chunk, pipe, threads, extra)
    R.close()
    G.close()


def test_chunks_beyond_16384_reads_in_both_placements(monkeypatch):
    """chunk sizes 17 216 / 32 768 / 65 536 with graph chaining on the host and on the device.  (The round-2 fault was a job whose FIRST
    mini-batch is one chunk -- it ran inline on pipeline context 0 while the next batch started worker 0 on the same context; any MGA_CHUNK with
    chunk/4 >= the 6 400 reads of the first 64 Mbp batch triggered it.)  70 000 x 3 kb reads; the default-chunk output is the yardstick, and that
    configuration is compared with the reference in the other tests of this file."""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "20000000", "-c", "2", "-H", "3", "-n", "70000", "-l", "3000", "-s", "9"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
    m = mga.map_files_idx(G, [reads], n_threads=8)
    want = hashlib.md5(m.view().tobytes()).hexdigest()
    n_want = len(m)
    m.free()
    assert n_want > 70000 * 2000
    L = mga.load()
    for chunk in (17216, 32768, 65536):
        for dev in ("0", "1"):
            monkeypatch.setenv("MGA_CHUNK", str(chunk))
            monkeypatch.setenv("MGA_DEV_GCHAIN", dev)
            L.mga_idx_stream_close(G.gi)   # the index's pipeline re-reads its knobs when it is rebuilt
            m = mga.map_files_idx(G, [reads], n_threads=8)
            got = hashlib.md5(m.view().tobytes()).hexdigest()
            m.free()
            assert got == want, (chunk, dev)
    G.close()


def test_long_reads_whose_first_batch_is_one_chunk_vs_reference_binary(monkeypatch):
    """50 kb reads.  The first mini-batch of a job is 64 Mbp = 1 280 such reads = ONE chunk at any chunk size >= 5 120, the second
    (500 Mbp, 10 000 reads) is several: the shape that made two threads share a pipeline context in round 2.  Default chunking (bounded by
    bases here) and MGA_CHUNK=6000, both placements of graph chaining, against the reference binary."""
    need_ref()
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "30000000", "-c", "2", "-H", "3", "-n", "11300", "-l", "50000", "-s", "13"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    ref_out = os.path.join(d, "ref.gaf")
    run_ref(["-c", "-x", "lr", "-t", "32", graph, reads], ref_out)
    want = hashlib.md5(open(ref_out, "rb").read()).hexdigest()
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
    L = mga.load()
    for chunk in ("", "6000"):
        for dev in ("0", "1"):
            if chunk:
                monkeypatch.setenv("MGA_CHUNK", chunk)
            monkeypatch.setenv("MGA_DEV_GCHAIN", dev)
            L.mga_idx_stream_close(G.gi)
            m = mga.map_files_idx(G, [reads], n_threads=8)
            got = hashlib.md5(m.view().tobytes()).hexdigest()
            m.free()
            assert got == want, (chunk, dev)
    G.close()


def test_device_chaining_gives_up_gracefully(monkeypatch):
    """a read that outgrows the scratch of the first launch is run again in the large arena; one that outgrows that too (here: both arenas tiny) is
    handed to the host threads instead of failing the job -- same bytes as the all-host and the normal device placement"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "6000000", "-c", "2", "-H", "4", "-n", "1200", "-s", "23"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
    L = mga.load()
    out = {}
    for tag, env in (("host", {"MGA_DEV_GCHAIN": "0"}), ("device", {"MGA_DEV_GCHAIN": "1"}),
                     ("retry", {"MGA_DEV_GCHAIN": "1", "MGA_GC_ARENA_KB": "48"}), ("give_up", {"MGA_DEV_GCHAIN": "1", "MGA_GC_ARENA_KB": "40", "MGA_GC_ARENA1_KB": "56"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        L.mga_idx_stream_close(G.gi)
        mga.get_stats(G, reset=True)
        m = mga.map_files_idx(G, [reads], n_threads=8)
        out[tag] = (hashlib.md5(m.view().tobytes()).hexdigest(), mga.get_stats(G)["n_gc_retry"])
        m.free()
        for k in env:
            monkeypatch.delenv(k)
    G.close()
    assert out["host"][0] == out["device"][0] == out["retry"][0] == out["give_up"][0], out
    assert out["retry"][1] > 0 and out["give_up"][1] > 0, out   # the small arenas did send reads through the retry launch


def test_graph_image_load_maps_byte_identically_to_build():
    """SURVEY 8 f4 / VERDICT r2 #6: load(save(graph)) == build(graph): same occurrence thresholds (mg_opt_update sees the same index), same GAF bytes with graph
    chaining on the host (reads gi->g / gi->es: the mapped file and the reverse-complement block) and on the device, and the reference's bytes"""
    need_ref()
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "8000000", "-c", "3", "-H", "4", "-n", "1500", "-s", "17"], stderr=subprocess.DEVNULL)
    graph, reads, img = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa"), os.path.join(d, "t.mgi")
    ref_out = os.path.join(d, "ref.gaf")
    run_ref(["-c", "-x", "lr", "-t", "8", graph, reads], ref_out)
    want = open(ref_out, "rb").read()
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
    built = (G.mo.occ_max1, G.mo.lc_max_occ) if hasattr(G.mo, "lc_max_occ") else (G.mo.occ_max1,)
    m = mga.map_files_idx(G, [reads], n_threads=8)
    assert m.bytes() == want
    m.free()
    G.save_image(img)
    G.close()
    for dev in ("0", "1"):
        os.environ["MGA_DEV_GCHAIN"] = dev
        try:
            G2 = mga.Graph(img, preset="lr", cigar=True, n_threads=8, image=True)
            loaded = (G2.mo.occ_max1, G2.mo.lc_max_occ) if hasattr(G2.mo, "lc_max_occ") else (G2.mo.occ_max1,)
            assert loaded == built
            m = mga.map_files_idx(G2, [reads], n_threads=8)
            assert m.bytes() == want, dev
            m.free()
            G2.close()
        finally:
            del os.environ["MGA_DEV_GCHAIN"]


@pytest.mark.parametrize("ranks,reads", [(2, 3000), (4, 1500)])
def test_bench_launches_its_own_ranks(ranks, reads):
    """(4 ranks: round 5 -- the size-exact point-to-point gather with three senders into rank 0)
    `python bench.py --gpus 2` without a launcher around it starts two ranks itself (torch.distributed.run), shards ONE read file over them, gathers the
    GAF to rank 0 and reports n_gpus = 2 with the gathered text byte-identical to the reference (gloo: two ranks share the one GPU of the test box)"""
    import json
    import sys
    need_ref()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--backend", "gloo", "--reads", str(reads), "--genome", "30000000", "--chr", "2",
                        "--steps", "1", "--warmup", "1", "--cpu-reads", "6000", "--resident-steps", "0", "--one-placement", "--threads", "4"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == ranks and d["value"] > 0
    assert "byte-identical" in d.get("parity", ""), d.get("parity")
    assert "(6000 in all, ONE file)" in d["config"]["workload"]
    if ranks != 2:
        return
    # the plain launcher contract still holds: a WORLD_SIZE that disagrees with --gpus is refused, not silently overridden
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1"], env=env2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p2.returncode != 0 and b"n_gpus" not in p2.stdout


def test_reads_with_a_gap_beyond_the_exact_wfa_cap_vs_reference_binary():
    """reads whose middle 9-12 kb are 30-40 % diverged: the anchor gap there passes 1e8 WFA cells and the reference falls back to
    mwf_wfa_chain() (miniwfa.c:829-832); same bytes expected, incl. the read whose gap takes the D+I shortcut"""
    import numpy as np
    need_ref()
    rng = np.random.default_rng(11)

    def rnd(n):
        return bytes(rng.choice(list(b"ACGT"), n).tolist())

    def mut(s, sub, indel):
        out = bytearray()
        for c in s:
            r = rng.random()
            if r < sub:
                out.append(int(rng.choice([x for x in b"ACGT" if x != c])))
            elif r < sub + indel / 2:
                continue
            elif r < sub + indel:
                out.append(c)
                out.append(int(rng.choice(list(b"ACGT"))))
            else:
                out.append(c)
        return bytes(out)

    d = tempfile.mkdtemp()
    ref = rnd(60000)
    graph, reads = os.path.join(d, "g.fa"), os.path.join(d, "r.fa")
    open(graph, "wb").write(b">chr\n" + ref + b"\n")
    with open(reads, "wb") as f:
        for i, (n, sub, ind) in enumerate([(9000, 0.3, 0.1), (12000, 0.3, 0.1), (9000, 0.25, 0.05)]):
            st = 3000 + 1000 * i
            r = ref[st:st + 6000] + mut(ref[st + 6000:st + 6000 + n], sub, ind) + ref[st + 6000 + n:st + 12000 + n]
            f.write(b">r%d\n%s\n" % (i, r))
        f.write(b">plain\n%s\n" % ref[40000:50000])
    ref_out, got = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf")
    run_ref(["-c", "-x", "lr", "-t", "2", graph, reads], ref_out)
    mga.map_files(graph, [reads], got, cigar=True)
    if open(ref_out, "rb").read() != open(got, "rb").read():
        raise AssertionError(first_diff(ref_out, got))
    lines = open(got, "rb").read().split(b"\n")
    assert len(lines) == 5 and int(lines[1].split(b"\t")[10]) > 30000  # r1: block length = both sides of the unrelated stretch


def _np_mutate(rng, s, err):
    """numpy ONT-like errors: 40 % substitutions, 30 % insertions, 30 % deletions of `err`"""
    import numpy as np
    a = np.frombuffer(s, dtype=np.uint8)
    u = rng.random(len(a))
    out = []
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    for ch, x in zip(a.tolist(), u.tolist()):
        if x < err * 0.4:
            out.append(int(rng.choice(alpha[alpha != ch])))
        elif x < err * 0.7:
            out.append(ch)
            out.append(int(rng.choice(alpha)))
        elif x < err:
            continue
        else:
            out.append(ch)
    return bytes(out)


def _repeat_workload(d):
    """two chromosomes, the second carrying a 2 %-diverged 40 kb copy of a stretch of the first: secondary chains, mapq < 60"""
    import numpy as np
    rng = np.random.default_rng(23)

    def rnd(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes())

    comp = bytes.maketrans(b"ACGT", b"TGCA")
    a = rnd(200000)
    b = rnd(100000) + _np_mutate(rng, a[50000:90000], 0.02) + rnd(50000)
    graph, reads = os.path.join(d, "rep.fa"), os.path.join(d, "rep.reads.fa")
    open(graph, "wb").write(b">chrA\n" + a + b"\n>chrB\n" + b + b"\n")
    with open(reads, "wb") as f:
        for i in range(40):
            src = a if i % 2 == 0 else b
            lo = 45000 if i % 2 == 0 else 95000
            st = int(rng.integers(lo, lo + 40000))
            r = _np_mutate(rng, src[st:st + 8000], 0.08)
            if i % 3 == 0:
                r = r.translate(comp)[::-1]
            f.write(b">q%d\n%s\n" % (i, r))
        f.write(b">junk\n%s\n" % rnd(3000))
    return graph, reads


OPTION_SETS = [
    # (tag, reference command line, idx_opt, map_opt, flags)
    ("kw_bw_gap", ["-k", "15", "-w", "8", "-r", "300,10000", "-g", "3000", "-n", "3,3", "-m", "30,30"],
     dict(k=15, w=8), dict(bw=300, bw_long=10000, max_gap=3000, min_gc_cnt=3, min_lc_cnt=3, min_gc_score=30, min_lc_score=30), 0),
    ("occ_2nd_vc", ["-f", "0.001", "-U", "20,100", "-p", "0.5", "-N", "3", "--secondary=yes", "--show-unmap=yes", "--vc"],
     None, dict(occ_max1_frac=0.001, occ_max1=20, occ_max1_cap=100, pri_ratio=0.5, best_n=3), 0x2000 | 0x100000 | 0x800),
    ("lchain_out", ["-S", "--no-comp-path", "-j", "0.2", "-M", "0.3"], None, dict(div=0.2, mask_level=0.3), 0x800000 | 0x200000),
    ("chain_pens", ["--gap-pen", "0.5", "--max-lc-skip", "10", "--max-lc-iter", "1000", "--gdp-max-ed", "2000", "--max-gc-skip", "10",
                    "--max-gap-pre", "500", "--ref-bonus", "5"],
     None, dict(chn_pen_gap=0.5, max_lc_skip=10, max_lc_iter=1000, gdp_max_ed=2000, max_gc_skip=10, max_gap_pre=500, ref_bonus=5), 0),
    ("rmq_primary", ["--rmq=yes"], None, None, 0x8000),
    ("write_mz", ["--write-mz"], None, None, 0x1000000 | 0x800000),
    ("frag_len", ["-F", "40000"], None, dict(max_frag_len=40000), 0),
    # the forms of the path column on the device's GAF writer (k_gaf.hip; the sets with -S / --write-mz above take the host's writer): vertices by name, no compact
    # form, secondary chains + the lines of unmapped reads with the stable-sequence intervals
    ("vc", ["--vc"], None, None, 0x800),
    ("no_comp_path", ["--no-comp-path"], None, None, 0x200000),
    ("2nd_unmap", ["--secondary=yes", "--show-unmap=yes", "-p", "0.5", "-N", "3"], None, dict(pri_ratio=0.5, best_n=3), 0x2000 | 0x100000),     # per-read reference gap max(F - qlen, max_gap) in the DP (map-algo.c:383-386)
]


@pytest.mark.parametrize("workload", ["bubbles", "repeat"])
@pytest.mark.parametrize("tag,cli,idx_opt,map_opt,flags", OPTION_SETS, ids=[o[0] for o in OPTION_SETS])
def test_command_line_options_vs_reference_binary(workload, tag, cli, idx_opt, map_opt, flags, monkeypatch):
    # graph chaining is placed by the host threads a rank has (device when <= 12): every option set is run through BOTH placements,
    # the device one (k_gchain + k_plan) on the bubble graph, the host instantiation of the same routine on the repeat workload
    monkeypatch.setenv("MGA_DEV_GCHAIN", "1" if workload == "bubbles" else "0")
    """every mapping option of the reference's command line (main.c:131-216) set through mg_idxopt_t / mg_mapopt_t: same bytes"""
    need_ref()
    d = tempfile.mkdtemp()
    if workload == "bubbles":
        subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "1500000", "-H", "3", "-n", "150", "-s", "7"], stderr=subprocess.DEVNULL)
        graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    else:
        graph, reads = _repeat_workload(d)
    ref_out, got = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf")
    run_ref(["-c", "-x", "lr", "-t", "4"] + cli + [graph, reads], ref_out)
    mga.map_files(graph, [reads], got, idx_opt=idx_opt, map_opt=map_opt, flags=flags)
    if open(ref_out, "rb").read() != open(got, "rb").read():
        raise AssertionError(first_diff(ref_out, got))
    if workload == "repeat" and tag == "occ_2nd_vc":
        body = open(got, "rb").read()
        assert b"tp:A:S" in body and b"junk\t3000\t0\t0\t*" in body  # secondaries and the unmapped read were really printed


def test_several_query_files_and_small_minibatches_vs_reference_binary():
    """minigraph graph a.fa b.fq.gz with -K 300k: the files are mapped one after the other (gmap.c:203-208), each in mini-batches of
    300 kbp (reader thread -> mapper -> writer thread, output buffers swapped between them); same bytes, same order"""
    import gzip
    need_ref()
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "2000000", "-H", "3", "-n", "240", "-s", "77"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    recs = open(reads, "rb").read().split(b">")[1:]
    fa, fq = os.path.join(d, "a.fa"), os.path.join(d, "b.fq.gz")
    with open(fa, "wb") as f:
        for r in recs[:150]:
            f.write(b">" + r)
    with gzip.open(fq, "wb") as f:
        for r in recs[150:]:
            name, seq = r.split(b"\n", 1)
            seq = seq.replace(b"\n", b"")
            f.write(b"@" + name + b" some comment\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n")
    ref_out, got = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf")
    run_ref(["-c", "-x", "lr", "-t", "4", "-K", "300k", graph, fa, fq], ref_out)
    mga.map_files(graph, [fa, fq], got, map_opt=dict(mini_batch_size=300000))
    if open(ref_out, "rb").read() != open(got, "rb").read():
        raise AssertionError(first_diff(ref_out, got))
    assert open(got, "rb").read().count(b"\n") >= 240


def test_half_gigabase_graph_20k_reads_with_batch_seams(monkeypatch):
    """scale: 0.56 Gbp 5-haplotype graph in 8 chromosomes (2^28-slot table, 32-bit list offsets in use), 20 000 x 10 kb
    reads through mga_map_files_to_path with -K 30M (7 mini-batch seams inside ONE chunk pipeline, parallel FASTA reader), against the
    reference binary; then the same job through the sequential reader"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "440000000", "-c", "8", "-H", "5", "-n", "20000", "-s", "11"], stderr=subprocess.DEVNULL)
    os.remove(os.path.join(d, "t.lin.fa"))
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    ref_out, got = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf")
    run_ref(["-c", "-x", "lr", "-t", "16", graph, reads], ref_out)
    mga.map_files(graph, [reads], got, n_threads=16, map_opt=dict(mini_batch_size=30000000))
    if subprocess.call(["cmp", "-s", ref_out, got]) != 0:
        raise AssertionError(first_diff(ref_out, got))
    monkeypatch.setenv("MGA_NO_FAST_READER", "1")
    mga.map_files(graph, [reads], got, n_threads=16, map_opt=dict(mini_batch_size=70000000))
    assert subprocess.call(["cmp", "-s", ref_out, got]) == 0


def test_one_input_four_shards_one_gaf():
    """mga_map_files_shard: the ranks' outputs of one FASTA file (byte ranges) and of its gzip copy (slices of every mini-batch),
    assembled per segment in rank order, are the single-process bytes and the reference's"""
    import gzip
    from minigraph_amd.dist import assemble_segments
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "6000000", "-H", "3", "-n", "3001", "-s", "12"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    ref_out = os.path.join(d, "ref.gaf")
    run_ref(["-c", "-x", "lr", "-t", "8", graph, reads], ref_out)
    want = open(ref_out, "rb").read()
    open(reads + ".gz", "wb").write(gzip.compress(open(reads, "rb").read(), 1))
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
    G.mo.mini_batch_size = 4000000
    one = mga.map_files_idx(G, [reads], n_threads=8)
    assert one.bytes() == want and len(one.seg_len) == 1
    for path, world in ((reads, 4), (reads + ".gz", 3), (reads, 7)):
        parts = [mga.map_files_idx(G, [path], n_threads=8, rank=r, world=world) for r in range(world)]
        assert all(len(p) > 0 for p in parts)
        if path.endswith(".gz"):
            assert len(parts[0].seg_len) > 3   # every mini-batch is a segment
        assert assemble_segments([p.bytes() for p in parts], [p.seg_len for p in parts]) == want, (path, world)
    G.close()


def test_mg_map_from_several_threads_with_own_tbufs():
    """the reference's threading contract (gmap.c:84-99): one mg_tbuf_t per worker thread, mg_map() called concurrently against one
    shared index.  A small C harness (pthreads, no Python in the timed part) maps every read from 6 threads, each with its own
    mg_tbuf_t, through the public API of libminigraph_amd.so; per-read GAF must equal the serial run's and the reference's"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "3000000", "-H", "3", "-n", "360", "-l", "6000", "-s", "14"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    src = os.path.join(d, "mt.c")
    open(src, "w").write(r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "minigraph_amd.h"
typedef struct { const mg_idx_t *gi; const mg_mapopt_t *opt; int n, tid, nt; char **name, **seq; int *len; kstring_t *out; const gfa_t *g; } job_t;
static void *worker(void *a) {
	job_t *j = (job_t*)a; int i;
	mg_tbuf_t *b = mg_tbuf_init();
	for (i = j->tid; i < j->n; i += j->nt) {
		mg_gchains_t *gc = mg_map(j->gi, j->len[i], j->seq[i], b, j->opt, j->name[i]);
		int32_t ql = j->len[i];
		mg_write_gaf(&j->out[i], j->g, gc, 1, &ql, j->name[i], j->opt->flag, 0);
		mg_gchain_free(gc);
	}
	mg_tbuf_destroy(b);
	return 0;
}
int main(int argc, char **argv) {
	int nt = atoi(argv[3]), n = 0, m = 0, i; char line[1 << 16];
	char **name = 0, **seq = 0; int *len = 0;
	FILE *fp = fopen(argv[2], "r");
	mg_idxopt_t io; mg_mapopt_t mo; mg_ggopt_t go;
	mg_verbose = 1;
	mg_opt_set(0, &io, &mo, &go); mg_opt_set("lr", &io, &mo, &go); mo.flag |= MG_M_CIGAR;
	while (fgets(line, sizeof line, fp)) {
		line[strcspn(line, "\r\n")] = 0;
		if (line[0] == '>') { if (n == m) { m = m ? m * 2 : 256; name = realloc(name, m * sizeof *name); seq = realloc(seq, m * sizeof *seq); len = realloc(len, m * sizeof *len); } name[n] = strdup(line + 1); seq[n] = strdup(""); len[n++] = 0; }
		else { int l = strlen(line); seq[n-1] = realloc(seq[n-1], len[n-1] + l + 1); memcpy(seq[n-1] + len[n-1], line, l + 1); len[n-1] += l; }
	}
	gfa_t *g = gfa_read(argv[1]);
	mg_idx_t *gi = mg_index(g, &io, 4, &mo);
	kstring_t *out = calloc(n, sizeof *out);
	pthread_t th[64]; job_t job[64];
	for (i = 0; i < nt; ++i) { job_t J = { gi, &mo, n, i, nt, name, seq, len, out, g }; job[i] = J; pthread_create(&th[i], 0, worker, &job[i]); }
	for (i = 0; i < nt; ++i) pthread_join(th[i], 0);
	for (i = 0; i < n; ++i) if (out[i].l) fwrite(out[i].s, 1, out[i].l, stdout);
	mg_idx_destroy(gi); gfa_destroy(g);
	return 0;
}
''')
    exe = os.path.join(d, "mt")
    subprocess.check_call(["gcc", "-O1", "-I" + os.path.join(mga.ROOT, "include"), src, "-o", exe, mga.LIB_PATH, "-lpthread",
                           "-Wl,-rpath," + os.path.dirname(mga.LIB_PATH)])
    ref_out = os.path.join(d, "ref.gaf")
    run_ref(["-c", "-x", "lr", "-t", "4", graph, reads], ref_out)
    want = open(ref_out, "rb").read()
    for nt in (1, 6):
        got = subprocess.run([exe, graph, reads, str(nt)], stdout=subprocess.PIPE, check=True, timeout=600).stdout
        assert got == want, nt


DROPIN = os.path.join(os.path.dirname(rb.REF_BIN), "minigraph_dropin")
DROPIN_UNPATCHED = os.path.join(os.path.dirname(rb.REF_BIN), "minigraph_dropin_unpatched")


def run_bin(exe, args, out):
    with open(out, "wb") as fo:
        subprocess.check_call([exe] + args, stdout=fo, stderr=subprocess.DEVNULL, timeout=900)


def test_link_level_dropin_reference_front_end_on_this_library():
    """INTEGRATION.md 1a/1b as an executable fact: the reference's own main.c / gmap.c / ggen.c / asm-call.c / ggsimple.c ... compiled from
    the reference sources and LINKED AGAINST libminigraph_amd.so in place of its mapping-path objects (oracle/Makefile, target dropin).
    patched = the kt_for(worker_for) lines replaced by mg_map_batch(); unpatched = mg_map() per read from kt_for workers, one mg_tbuf_t each."""
    assert os.path.exists(DROPIN) and os.path.exists(DROPIN_UNPATCHED), "oracle/_ref/minigraph_dropin missing: make -C oracle dropin (needs /root/reference)"
    need_ref()
    d = tempfile.mkdtemp()
    mt, orang, chimp = (os.path.join(GOLD, f) for f in ("MT.gfa", "MT-orangA.fa", "MT-chimp.fa"))
    # (1) the reference's known answer through its own CLI and its own kt_pipeline, mapping on the GPU
    for exe, th in ((DROPIN, "4"), (DROPIN_UNPATCHED, "4"), (DROPIN_UNPATCHED, "1")):
        out = os.path.join(d, "mt.gaf")
        run_bin(exe, ["-cx", "lr", "-t", th, mt, orang], out)
        assert hashlib.md5(open(out, "rb").read()).hexdigest() == "22bf23ebe2039e8353f56f4a324a2eaa", (exe, th)
    # (2) synthetic bubble graph, 300 reads, both binaries against the reference binary
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "2000000", "-H", "3", "-n", "300", "-s", "8"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    ref_out = os.path.join(d, "ref.gaf")
    run_ref(["-cx", "lr", "-t", "4", graph, reads], ref_out)
    for exe in (DROPIN, DROPIN_UNPATCHED):
        out = os.path.join(d, "syn.gaf")
        run_bin(exe, ["-cx", "lr", "-t", "6", graph, reads], out)
        if open(out, "rb").read() != open(ref_out, "rb").read():
            raise AssertionError(exe + ": " + first_diff(ref_out, out))


@pytest.mark.parametrize("args", [["-cxasm", "--call"], ["-cxggs"], ["-cxasm", "--cov"]])
def test_dropin_call_and_graph_generation_consume_our_chains(args):
    """`--call` (ggen.c:128-139 -> mg_call_asm, asm-call.c:21), incremental graph generation (`-x ggs`, ggsimple.c) and `--cov`
    are the reference's own code consuming the mg_gchains_t objects THIS library returns from mg_map_batch() (chains, lc[], anchors a[],
    CIGARs): their outputs must equal the all-reference binary's byte for byte -- parity of the in-memory results, not just of GAF text"""
    assert os.path.exists(DROPIN), "oracle/_ref/minigraph_dropin missing: make -C oracle dropin"
    need_ref()
    d = tempfile.mkdtemp()
    # the reference's own fixtures, then a bubble graph with eight 300 kb contigs (0.4 % divergence) that walk through ~100 bubbles
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "1500000", "-H", "3", "-n", "8", "-l", "300000", "-e", "0.004", "-s", "19"], stderr=subprocess.DEVNULL)
    jobs = [(os.path.join(GOLD, "MT.gfa"), os.path.join(GOLD, q)) for q in ("MT-orangA.fa", "MT-chimp.fa")] + [(os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa"))]
    total = 0
    for graph, q in jobs:
        want, got = os.path.join(d, "want.out"), os.path.join(d, "got.out")
        run_bin(rb.REF_BIN, args + ["-t", "4", graph, q], want)
        run_bin(DROPIN, args + ["-t", "4", graph, q], got)
        a, b = open(want, "rb").read(), open(got, "rb").read()
        assert a == b, (args, q, len(a), len(b))
        total += len(a)
    assert total > 1000, total


@pytest.mark.parametrize("workload", ["bubbles5", "repeat"])
def test_graph_chaining_on_device_equals_host_instantiation_and_reference(monkeypatch, workload):
    """k_gchain (gc_core.h on one GPU lane per read: chain records, clean-up, DP + shortest walks, GWFA bridging, ordering, parents, filters)
    against the SAME routine on host threads (MGA_HOST_GCHAIN=1) and against the reference binary; the device path must really have run
    (GWFA bridges and shortest-walk searches counted by the kernel)"""
    need_ref()
    d = tempfile.mkdtemp()
    if workload == "bubbles5":
        subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "8000000", "-H", "5", "-n", "3000", "-l", "12000", "-s", "61"], stderr=subprocess.DEVNULL)
        graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    else:
        graph, reads = _repeat_workload(d)
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
    R = mga.Reads(reads)
    mga.get_stats(G, reset=True)
    monkeypatch.setenv("MGA_DEV_GCHAIN", "1")
    dev = mga.map_reads(G, R, n_threads=8)
    st = mga.get_stats(G, reset=True)
    if workload == "bubbles5":
        assert st["n_gwfa"] > 500 and st["n_shortk"] > 500, st
    assert st["n_gc_retry"] == 0 and 0 < st["gc_arena_peak"] < (1 << 20), st
    assert st["n_wfa_dev_plan"] > 0.99 * st["n_wfa"] > 0, st   # the gap list was made on the device too (k_plan.hip; all but the reads the host chained)
    mga.prof_enable(True); mga.prof_get(reset=True)
    monkeypatch.setenv("MGA_GC_SPLIT", "1")                     # the three-launch form (k_gchain_p1 per read / k_gchain_p2 per bridge / k_gchain_p3 per read): same bytes
    dev_split = mga.map_reads(G, R, n_threads=8)
    pr = mga.prof_get(reset=True); mga.prof_enable(False)
    assert pr["k_gchain_p2"][1] > 0 and pr["k_gchain_p3"][1] > 0, pr   # ... and it did run
    assert dev_split == dev
    monkeypatch.delenv("MGA_GC_SPLIT"); mga.get_stats(G, reset=True)
    monkeypatch.setenv("MGA_DEV_PLAN", "0")                     # device chains, gap list by host threads (align.c)
    dev_hostplan = mga.map_reads(G, R, n_threads=8)
    st1 = mga.get_stats(G, reset=True)
    assert st1["n_wfa_dev_plan"] == 0 and st1["n_wfa"] == st["n_wfa"] and st1["wfa_t_bases"] == st["wfa_t_bases"] and st1["wfa_q_bases"] == st["wfa_q_bases"], (st, st1)
    assert dev_hostplan == dev
    monkeypatch.delenv("MGA_DEV_PLAN")
    monkeypatch.setenv("MGA_DEV_GCHAIN", "0")
    host = mga.map_reads(G, R, n_threads=8)
    st2 = mga.get_stats(G, reset=True)
    assert st2["n_gwfa"] == 0          # nothing was counted by the kernel on the host pass
    R.close()
    G.close()
    assert dev == host
    ref_out = os.path.join(d, "ref.gaf")
    run_ref(["-c", "-x", "lr", "-t", "8", graph, reads], ref_out)
    assert open(ref_out, "rb").read() == dev


def test_text_pool_is_regrown_and_relaunched(monkeypatch):
    """the device text pool is sized for ordinary reads; when a chunk needs more (very divergent reads, many printed secondaries) the kernel
    reports what it would have written and the stage runs again with a pool of that size -- forced here with a tiny pool"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "2000000", "-H", "3", "-n", "500", "-s", "71"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=8)
    R = mga.Reads(reads)
    want = mga.map_reads(G, R, n_threads=8)
    monkeypatch.setenv("MGA_TXT_TIGHT", "1")
    got = mga.map_reads(G, R, n_threads=8)
    R.close()
    G.close()
    assert got == want and want.count(b"\tcg:Z:") >= 490


def test_device_placement_is_deterministic_over_repeated_runs(monkeypatch):
    """the all-device placement (k_gchain -- and, in the second sweep, its three-launch form k_gchain_p1 / p2 / p3 --, k_plan, the windowed WFA ladder with its device-side work lists and atomically
    reserved pools, k_text) maps ONE workload 20 times: every run gives the same bytes, and they are the reference's.  Atomics only hand out PLACES (pool offsets, list
    slots, job order); nothing a place decides may reach the output.  A second sweep runs with few resident wavefronts and small job quanta so that the order in which
    reads / bridges / gaps are picked up differs from the first."""
    need_ref()
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "12000000", "-c", "2", "-H", "5", "-n", "4000", "-s", "71"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    ref_out = os.path.join(d, "ref.gaf")
    run_ref(["-c", "-x", "lr", "-t", "8", graph, reads], ref_out)
    want = hashlib.md5(open(ref_out, "rb").read()).hexdigest()
    monkeypatch.setenv("MGA_DEV_GCHAIN", "1")
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=4)
    seen = set()
    for it in range(20):
        m = mga.map_files_idx(G, [reads], n_threads=4)
        seen.add(hashlib.md5(m.view().tobytes()).hexdigest())
        m.free()
    assert seen == {want}, seen
    st = mga.get_stats(G)
    assert st["n_gwfa"] > 0 and st["n_wfa_dev_plan"] > 0   # the device did chain and plan
    G.close()
    monkeypatch.setenv("MGA_GC_WAVES", "96")     # other pick-up orders: the three-launch form with 96 resident wavefronts per part, chunks of 1000 reads, two chunks in flight
    monkeypatch.setenv("MGA_GC_SPLIT", "1")
    monkeypatch.setenv("MGA_GC_WAVES2", "160")
    monkeypatch.setenv("MGA_CHUNK", "1000")
    code = ("import sys, hashlib; sys.path.insert(0, %r); import minigraph_amd as mga\n"
            "G = mga.Graph(sys.argv[1], preset='lr', cigar=True, n_threads=4)\n"
            "s = set()\n"
            "for it in range(6):\n"
            "    m = mga.map_files_idx(G, [sys.argv[2]], n_threads=4); s.add(hashlib.md5(m.view().tobytes()).hexdigest()); m.free()\n"
            "print('SEEN', ' '.join(sorted(s)))\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    p = subprocess.run([sys.executable, "-c", code, graph, reads], env=dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)   # (a child: the knobs are read once)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("SEEN")][0].split()[1:]
    assert line == [want], line


def test_asm_200Mbp_four_contigs_gaf_and_sharded_call_vs_reference_binary():
    """BASELINE configs[4] at a scale the test box carries: four 50 Mbp contigs (200 Mbp of query) against a 200 Mbp 3-haplotype graph in 4 chromosomes,
    `-cx asm`: (1) file -> file GAF = the reference binary's bytes; (2) the contigs as two SHARDS (what two ranks of a node would take), each mapped to mg_gchains_t by
    mg_map_batch() on the GPU, packed, unpacked on "rank 0" (mga_gchains_pack / _unpack: what dist.gather_chains moves), and fed in input order to the REFERENCE's own
    mg_call_asm (asm-call.c:21) = the BED of `minigraph -cxasm --call`."""
    import ctypes as C
    need_ref()
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "200000000", "-c", "4", "-H", "3", "-n", "4", "-l", "50000000", "-e", "0.001", "-s", "5"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    ref_out, got, ref_bed = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf"), os.path.join(d, "ref.bed")
    run_ref(["-c", "-x", "asm", "-t", "16", graph, reads], ref_out)
    mga.map_files(graph, [reads], got, preset="asm", cigar=True, n_threads=16)
    if open(ref_out, "rb").read() != open(got, "rb").read():
        raise AssertionError(first_diff(ref_out, got))
    assert os.path.getsize(got) > 1000000
    # ---- (2) ----
    run_ref(["-c", "-x", "asm", "--call", "-t", "16", graph, reads], ref_bed)
    import hostpipe as hp
    names, seqs = hp.read_fa(reads)
    L = mga.load()
    G = mga.Graph(graph, preset="asm", cigar=True, n_threads=16)
    # round 5: through the library's own entries -- mga_ggen_map_shard (a rank's part of ggen_map, ggen.c:64: its shard by BASES, mapped and packed) for each of three
    # "ranks", mga_ggen_assemble on "rank 0" (what dist.ggen_map_sharded does with an RCCL gather in between)
    from minigraph_amd.dist import ggen_assemble
    L.mga_ggen_map_shard.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.mga_gchains_unpack.restype = C.POINTER(C.c_void_p)
    L.mga_gchains_unpack.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_int)]
    L.mg_gchain_free.argtypes = [C.c_void_p]
    L.mga_free.argtypes = [C.c_void_p]
    n = len(names)
    qlens, sp, npp = (C.c_int * n)(*[len(x) for x in seqs]), (C.c_char_p * n)(*seqs), (C.c_char_p * n)(*names)
    parts, world = [], 3
    for rank in range(world):
        buf, nb = C.c_void_p(), C.c_int64(0)
        assert L.mga_ggen_map_shard(G.gi, n, qlens, sp, npp, C.byref(G.mo), 16, rank, world, C.byref(buf), C.byref(nb)) == 0, L.mga_last_error()
        parts.append(C.string_at(buf, nb.value))
        L.mga_free(buf)
    k = C.c_int(0)
    assert not L.mga_gchains_unpack(parts[0][:len(parts[0]) // 2], len(parts[0]) // 2, C.byref(k))   # a truncated buffer is refused, not read past
    gathered = ggen_assemble(parts, n)
    G.close()
    R = rb.Ref().lib

    class bseq1_t(C.Structure):  # mg_bseq1_t, bseq.h:14-17
        _fields_ = [("l_seq", C.c_int32), ("rid", C.c_int32), ("name", C.c_char_p), ("seq", C.c_char_p), ("qual", C.c_char_p), ("comment", C.c_char_p)]
    R.gfa_read.restype = C.c_void_p
    R.mg_call_asm.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    g = R.gfa_read(graph.encode())
    sq = (bseq1_t * len(names))()
    for i in range(len(names)):
        sq[i].l_seq, sq[i].rid, sq[i].name, sq[i].seq = len(seqs[i]), i, names[i], seqs[i]
    arr = (C.c_void_p * len(gathered))(*gathered)
    bed = os.path.join(d, "got.bed")
    libc = C.CDLL(None)
    libc.fflush(None)
    fd, keep = os.open(bed, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644), os.dup(1)
    os.dup2(fd, 1)
    try:
        R.mg_call_asm(g, len(names), sq, arr, 5, 100000)
        libc.fflush(None)
    finally:
        os.dup2(keep, 1)
        os.close(fd)
        os.close(keep)
    want = open(ref_bed, "rb").read()
    assert want.count(b"\n") > 1000
    assert open(bed, "rb").read() == want


@pytest.mark.parametrize("flags,cli", [(0, []), (0x200000, ["--no-comp-path"]), (0x800, ["--vc"])])
def test_gaf_lines_on_device_long_walks_mixed_tags_long_names(flags, cli):
    """round 6: whole GAF lines are written on the device (k_gaf.hip).  A graph of 120 bp segments -- a 8 kb read walks ~70 vertices: the path column is folded across more than
    one block of 64 vertices -- most of them intervals of ONE rank-0 stable sequence (SN / SO / SR tags: the compact one-interval form, forward and reverse), every 150th
    without tags (printed by name in the middle of a run of intervals), alternative segments of rank 1 on stable sequences of their own, and read names of 120 characters
    (copied in pieces, not through the 64-byte stage): the reference's bytes, in the general form, without the compact form and with vertex coordinates"""
    import numpy as np
    need_ref()
    rng = np.random.default_rng(61)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    L, S = 60000, 120
    back = bytes(rng.choice(acgt, L).tobytes())
    d = tempfile.mkdtemp()
    gfa, reads = os.path.join(d, "g.gfa"), os.path.join(d, "r.fa")
    n = L // S
    with open(gfa, "wb") as f:
        for i in range(n):
            seq = back[i * S:(i + 1) * S]
            if i % 150 == 149:
                f.write(b"S\ts%d\t%s\n" % (i, seq))   # no stable-sequence tags: a vertex printed by name
            else:
                f.write(b"S\ts%d\t%s\tSN:Z:chrK\tSO:i:%d\tSR:i:0\n" % (i, seq, i * S))
        for i in range(n - 1):
            f.write(b"L\ts%d\t+\ts%d\t+\t0M\n" % (i, i + 1))
        for i in range(5, n - 2, 10):   # a bubble around segment i + 1: an alternative allele of rank 1 on its own stable sequence
            alt = bytes(rng.choice(acgt, 90).tobytes())
            f.write(b"S\ta%d\t%s\tSN:Z:alt%d\tSO:i:0\tSR:i:1\n" % (i, alt, i))
            f.write(b"L\ts%d\t+\ta%d\t+\t0M\nL\ta%d\t+\ts%d\t+\t0M\n" % (i, i, i, i + 2))
    with open(reads, "wb") as f:
        for k in range(60):
            st = int(rng.integers(0, L - 8000))
            r = _np_mutate(rng, back[st:st + 8000], 0.05)
            if k % 2:
                r = r.translate(comp)[::-1]
            f.write(b">read_%03d_%s\n%s\n" % (k, b"x" * 108, r))
    ref_out, got = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf")
    run_ref(["-c", "-x", "lr", "-t", "4"] + cli + [gfa, reads], ref_out)
    mga.map_files(gfa, [reads], got, flags=flags)
    if open(ref_out, "rb").read() != open(got, "rb").read():
        raise AssertionError(first_diff(ref_out, got))
    body = open(got, "rb").read()
    assert body.count(b"\n") >= 50 and (flags != 0 or (b"\t-\tchrK\t" in body and b"\t+\tchrK\t" in body and (b">chrK:" in body or b"<chrK:" in body) and (b">s149" in body or b"<s149" in body or b">s299" in body or b"<s299" in body))), "the forms this test is about were not printed"
