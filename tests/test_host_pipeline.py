"""CPU-only parity of the product's host-side stages (RMQ rescue, lchain clean-up, graph chaining,
shortest-k, GWFA bridging, MAPQ, CIGAR stitching, ds, GAF) against the unmodified reference, with the
oracle standing in for the HIP kernels (tests/hostpipe.py)."""
import hashlib
import os
import subprocess
import tempfile

import pytest

import hostpipe as hp
import minigraph_amd as mga
import refbind as rb

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
pytestmark = pytest.mark.skipif(not (rb.have_oracle() and os.path.exists(mga.LIB_PATH)), reason="libraries not built")


def test_mt_known_answer_host_logic():
    """reference KAT: md5(minigraph -cx lr MT.gfa MT-orangA.fa) = 22bf23eb...  (occ_max1=50, lc_max_occ=2 on this graph)"""
    gaf, n_prob = hp.map_with_oracle_stages(os.path.join(GOLD, "MT.gfa"), os.path.join(GOLD, "MT-orangA.fa"), 50, 2)
    assert n_prob > 0
    assert hashlib.md5(gaf).hexdigest() == "22bf23ebe2039e8353f56f4a324a2eaa"


@pytest.mark.parametrize("query", ["MT-chimp.fa", "MT-human.fa"])
def test_mt_golden(query):
    gaf, _ = hp.map_with_oracle_stages(os.path.join(GOLD, "MT.gfa"), os.path.join(GOLD, query), 50, 2)
    assert gaf == open(os.path.join(GOLD, query.replace(".fa", ".cx_lr.gaf")), "rb").read()


@pytest.mark.skipif(not os.path.exists(rb.REF_BIN), reason="oracle/_ref/minigraph not built")
@pytest.mark.parametrize("cigar", [True, False])
def test_synthetic_graph_vs_reference_binary(cigar):
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "600000", "-H", "3", "-n", "60", "-s", "3"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    want, occ, lco = hp.run_reference(graph, reads, cigar=cigar)
    got, _ = hp.map_with_oracle_stages(graph, reads, occ, lco, cigar=cigar)
    assert got == want


@pytest.mark.skipif(not os.path.exists(rb.REF_BIN), reason="oracle/_ref/minigraph not built")
@pytest.mark.parametrize("cigar", [True, False])
def test_asm_preset_host_phases_vs_reference_binary(cigar):
    """-x asm: the RMQ chainer is the primary chainer and runs in the product's host phases (sorted anchors -> (segment, strand) runs ->
    forward passes on the thread pool -> backtracking -> rescue pass); 300 kb contigs give several runs per contig"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "1500000", "-H", "3", "-n", "6", "-l", "300000", "-e", "0.004", "-s", "9"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    want, occ, lco = hp.run_reference(graph, reads, cigar=cigar, preset="asm")
    got, _ = hp.map_with_oracle_stages(graph, reads, occ, lco, cigar=cigar, preset="asm")
    assert got == want
    if not cigar:   # the phases as a task graph over 1 / 3 / 7 threads (a read backtracks while the next one's forward runs are taken) and with barriers between them (MGA_RQ_BARRIERS=1)
        for nt, barriers in ((1, "0"), (3, "0"), (7, "0"), (5, "1")):
            os.environ["MGA_RQ_BARRIERS"] = barriers
            try:
                got2, _ = hp.map_with_oracle_stages(graph, reads, occ, lco, cigar=False, preset="asm", n_threads=nt)
            finally:
                del os.environ["MGA_RQ_BARRIERS"]
            assert got2 == want, (nt, barriers)


def _mutate(rng, s, rate):
    import numpy as np
    out = bytearray()
    for ch in s:
        u = rng.random()
        if u < rate * 0.4:
            out.append(int(rng.choice([c for c in b"ACGT" if c != ch])))
        elif u < rate * 0.7:
            out.append(int(rng.choice(list(b"ACGT"))))
            out.append(ch)
        elif u < rate:
            pass
        else:
            out.append(ch)
    return bytes(out)


@pytest.mark.skipif(not os.path.exists(rb.REF_BIN), reason="oracle/_ref/minigraph not built")
def test_graph_chaining_core_many_reads_five_haplotypes():
    """gc_core.h on the host (chain records, clean-up, DP + shortest walks, GWFA / walk bridging, ordering, parents, filters) over 500 reads
    that cross ~1.5 bubbles each on a 5-haplotype graph; no CIGAR, so the oracle's Python WFA loop does not dominate the test"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "3000000", "-H", "5", "-n", "500", "-l", "12000", "-s", "41"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    want, occ, lco = hp.run_reference(graph, reads, cigar=False)
    got, _ = hp.map_with_oracle_stages(graph, reads, occ, lco, cigar=False)
    assert got == want
    assert sum(1 for l in want.split(b"\n") if l.count(b">") + l.count(b"<") >= 3) > 50   # walks over several segments (through alt alleles: not compacted into one stable interval) were really bridged


@pytest.mark.skipif(not os.path.exists(rb.REF_BIN), reason="oracle/_ref/minigraph not built")
def test_graph_chaining_core_repeats_secondaries():
    """two chromosomes sharing a 2 %-diverged 40 kb stretch: several graph chains per read, parents / secondaries / sub-scores / MAPQ < 60"""
    import numpy as np
    rng = np.random.default_rng(23)
    d = tempfile.mkdtemp()
    rnd = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes())
    a = rnd(200000)
    b = rnd(100000) + _mutate(rng, a[50000:90000], 0.02) + rnd(50000)
    graph, reads = os.path.join(d, "rep.fa"), os.path.join(d, "rep.reads.fa")
    open(graph, "wb").write(b">chrA\n" + a + b"\n>chrB\n" + b + b"\n")
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    with open(reads, "wb") as f:
        for i in range(60):
            src, lo = (a, 45000) if i % 2 == 0 else (b, 95000)
            st = int(rng.integers(lo, lo + 40000))
            r = _mutate(rng, src[st:st + 8000], 0.08)
            if i % 3 == 0:
                r = r.translate(comp)[::-1]
            f.write(b">q%d\n%s\n" % (i, r))
    want, occ, lco = hp.run_reference(graph, reads, cigar=False)
    got, _ = hp.map_with_oracle_stages(graph, reads, occ, lco, cigar=False)
    assert got == want
    mapqs = [int(l.split(b"\t")[11]) for l in want.split(b"\n") if l]
    assert min(mapqs) < 60 and max(mapqs) == 60   # sub-optimal chains pulled some MAPQs down


@pytest.mark.skipif(not os.path.exists(rb.REF_BIN), reason="oracle/_ref/minigraph not built")
@pytest.mark.parametrize("level", ["1", "2", "3"])
def test_three_part_form_of_graph_chaining_bridges_in_reverse_order(level):
    """the device runs a read in three kernels (gc_read_p1 / a wavefront per bridge: gc_job_run / gc_read_p3, gc_core.h) because a read's bridges are independent of each other
    and of the assembly; MGA_GC_SPLIT_TEST=1 makes the HOST instantiation take the same three parts, the bridges last to first in an arena of their own: the same bytes as the
    reference binary (a child process: the switch is read once).  Level 3 also reports every bridge between neighbouring chains as "no walk of the chosen length", so that part 3
    takes its redo path (the pairs in between, computed where they are met) for all of them.  Level 2 is the device's hand-over (k_gchain_p1 -> p2 -> p3): part 1's state leaves
    through one block (the layout functions the kernels use), its arena and its anchor copy are scrubbed, the jobs and part 3 work from the block"""
    import sys
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "3000000", "-H", "5", "-n", "500", "-l", "12000", "-s", "43"], stderr=subprocess.DEVNULL)
    child = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\nimport hostpipe as hp\n"
             "g, r = sys.argv[1], sys.argv[2]\nwant, occ, lco = hp.run_reference(g, r, cigar=False)\ngot, _ = hp.map_with_oracle_stages(g, r, occ, lco, cigar=False)\n"
             "print('SAME', int(got == want), len(got))\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", child, os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")], env=dict(os.environ, MGA_GC_SPLIT_TEST=level),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("SAME")][0].split()
    assert line[1] == "1" and int(line[2]) > 10000


@pytest.mark.skipif(not os.path.exists(rb.REF_BIN), reason="oracle/_ref/minigraph not built")
def test_ultra_long_read_placement_first_pass_on_host_threads():
    """-x lr reads beyond MGA_LONG_READ bases are chained on host threads (hchain.c: mga_lchain_dp_fwd restates mg_lchain_dp's forward loop, lchain.c:168-207; the long-join
    rescue follows through the RMQ tree as for every read the device defers): the same bytes as the reference on reads with many chains, strays and rescues"""
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "2500000", "-H", "4", "-n", "12", "-l", "150000", "-e", "0.08", "-s", "77"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    want, occ, lco = hp.run_reference(graph, reads, cigar=False)
    got, _ = hp.map_with_oracle_stages(graph, reads, occ, lco, cigar=False, host_dp=True)
    assert got == want
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "u"), "-G", "1500000", "-H", "3", "-n", "300", "-l", "9000", "-s", "78"], stderr=subprocess.DEVNULL)   # ordinary reads through the same code
    graph, reads = os.path.join(d, "u.gfa"), os.path.join(d, "u.reads.fa")
    want, occ, lco = hp.run_reference(graph, reads, cigar=False)
    got, _ = hp.map_with_oracle_stages(graph, reads, occ, lco, cigar=False, host_dp=True)
    assert got == want
