"""End-to-end parity on the GPU: the whole `-cx lr` job through the C ABI, GAF bytes compared with
the unmodified reference (golden files, and the reference binary itself where oracle/_ref travelled)."""
import hashlib
import os
import subprocess
import tempfile

import pytest

import minigraph_amd as mga
import refbind as rb

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_ref(args, out):
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
    if not os.path.exists(rb.REF_BIN):
        pytest.skip("oracle/_ref/minigraph not present")
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "3000000", "-H", "3", "-n", "400", "-s", "5"], stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t." + target), os.path.join(d, "t.reads.fa")
    ref_out, got = os.path.join(d, "ref.gaf"), os.path.join(d, "got.gaf")
    run_ref((["-c"] if cigar else []) + ["-x", "lr", "-t", "4", graph, reads], ref_out)
    mga.map_files(graph, [reads], got, cigar=cigar)
    if open(ref_out, "rb").read() != open(got, "rb").read():
        raise AssertionError(first_diff(ref_out, got))
