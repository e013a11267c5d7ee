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
