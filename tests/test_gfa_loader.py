"""The library's rGFA / FASTA loader (csrc/gfa_load.c) against the reference's gfa_read + gfa_finalize (gfa-io.c:294-340, gfa-base.c:421-430):
the in-memory graph must be the SAME graph field by field -- segments, stable sequences, and above all the arc array in the same ORDER (the order of
the arcs leaving a vertex decides ties in mg_shortest_k and the GWFA), with the same overlaps, ranks, link ids and complement flags, and the same index.
CPU test: both loaders are host code; the reference side is oracle/_ref/libmgref.so (the unmodified reference)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import refbind as rb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


class gfa_aux_t(C.Structure):
    _fields_ = [("m_aux", C.c_uint32), ("l_aux", C.c_uint32), ("aux", C.c_void_p)]


class gfa_seg_t(C.Structure):
    _fields_ = [("len", C.c_int32), ("del_circ", C.c_uint32), ("snid", C.c_int32), ("soff", C.c_int32), ("rank", C.c_int32),
                ("name", C.c_char_p), ("seq", C.c_void_p), ("utg", C.c_void_p), ("aux", gfa_aux_t)]


class gfa_sseq_t(C.Structure):
    _fields_ = [("name", C.c_char_p), ("min", C.c_int32), ("max", C.c_int32), ("rank", C.c_int32)]


class gfa_t(C.Structure):
    _fields_ = [("m_seg", C.c_uint32), ("n_seg", C.c_uint32), ("max_rank", C.c_uint32), ("seg", C.POINTER(gfa_seg_t)), ("h_names", C.c_void_p),
                ("m_sseq", C.c_uint32), ("n_sseq", C.c_uint32), ("sseq", C.POINTER(gfa_sseq_t)), ("h_snames", C.c_void_p),
                ("m_arc", C.c_uint64), ("n_arc", C.c_uint64), ("arc", C.c_void_p), ("link_aux", C.c_void_p), ("idx", C.POINTER(C.c_uint64))]


arc_dt = np.dtype([("v_lv", "<u8"), ("w", "<u4"), ("rank", "<i4"), ("ov", "<i4"), ("ow", "<i4"), ("bits", "<u8")])   # gfa.h:33-39: link_id:61, strong:1, del:1, comp:1


def snapshot(lib, path):
    lib.gfa_read.restype = C.c_void_p
    lib.gfa_read.argtypes = [C.c_char_p]
    lib.gfa_destroy.argtypes = [C.c_void_p]
    p = lib.gfa_read(path.encode())
    assert p, "gfa_read failed on " + path
    g = C.cast(p, C.POINTER(gfa_t)).contents
    segs = []
    for i in range(g.n_seg):
        s = g.seg[i]
        seq = C.string_at(s.seq, s.len) if s.seq else None
        segs.append((s.len, s.del_circ & 0xffff, s.snid, s.soff, s.rank, s.name, seq))
    sseq = [(g.sseq[i].name, g.sseq[i].min, g.sseq[i].max, g.sseq[i].rank) for i in range(g.n_sseq)]
    arcs = np.empty(g.n_arc, dtype=arc_dt)
    if g.n_arc:
        C.memmove(arcs.ctypes.data, g.arc, int(g.n_arc) * 32)
    idx = np.array([g.idx[i] for i in range(2 * g.n_seg)], dtype=np.uint64)
    out = dict(n_seg=g.n_seg, max_rank=g.max_rank, segs=segs, sseq=sseq, arcs=arcs, idx=idx)
    lib.gfa_destroy(p)
    return out


def same_graph(a, b, what):
    assert a["n_seg"] == b["n_seg"] and a["max_rank"] == b["max_rank"], what
    assert a["segs"] == b["segs"], what + ": segments"
    assert a["sseq"] == b["sseq"], what + ": stable sequences"
    assert len(a["arcs"]) == len(b["arcs"]), what + ": arc count %d vs %d" % (len(a["arcs"]), len(b["arcs"]))
    for f in ("v_lv", "w", "rank", "ov", "ow", "bits"):
        bad = np.nonzero(a["arcs"][f] != b["arcs"][f])[0]
        assert len(bad) == 0, "%s: arc field %s differs first at arc %d: %r vs %r" % (what, f, bad[0], a["arcs"][bad[0]], b["arcs"][bad[0]])
    assert np.array_equal(a["idx"], b["idx"]), what + ": arc index"


def libs():
    if not rb.have_ref():
        pytest.skip("oracle/_ref/libmgref.so not built")
    import minigraph_amd as mga
    return mga.load(), C.CDLL(rb.REF_SO)


HAND = {
    # overlaps in every spelling gfa-io.c:216-245 accepts, incl. one-sided ':' forms that the complement line completes (gfa_fix_semi_arc) and ones nothing completes (arc dropped)
    "overlap_spellings": "\n".join([
        "S\ta\tACGTACGTAC", "S\tb\tGGGGGCCCCC", "S\tc\tTTTTTAAAAA", "S\td\tACACACACAC", "S\te\t*\tLN:i:25",
        "L\ta\t+\tb\t+\t3M", "L\tb\t+\tc\t-\t2M1D1M", "L\tc\t+\td\t+\t4:", "L\td\t-\tc\t-\t:4", "L\td\t+\te\t+\t2:3", "L\te\t+\ta\t+\t:", "L\ta\t-\te\t-\t*",
        "L\tb\t-\ta\t-\t3M", "L\te\t-\td\t-\t3:2"]) + "\n",
    # a link whose segment has no S-line (gfa_fix_no_seg), duplicate links, a link and its own complement both given, a self loop and a hairpin (v -> v^1: its own complement)
    "missing_dups_loops": "\n".join([
        "S\ts1\tACGTACGTACGT\tSN:Z:chr1\tSO:i:0\tSR:i:0", "S\ts2\tGGGG\tSN:Z:chr1\tSO:i:12\tSR:i:0", "S\ts3\tCCCCCC\tSN:Z:alt\tSO:i:12\tSR:i:1", "S\ts4\tTTTTTTTT\tSN:Z:chr1\tSO:i:16\tSR:i:0",
        "L\ts1\t+\ts2\t+\t0M\tSR:i:0", "L\ts1\t+\ts3\t+\t0M\tSR:i:1", "L\ts2\t+\ts4\t+\t0M\tSR:i:0", "L\ts3\t+\ts4\t+\t0M\tSR:i:1", "L\ts4\t-\ts3\t-\t0M\tSR:i:1",
        "L\ts1\t+\ts2\t+\t0M\tSR:i:0", "L\ts4\t+\tghost\t+\t0M", "L\ts4\t+\ts4\t+\t0M\tSR:i:2", "L\ts2\t+\ts2\t-\t0M\tSR:i:3", "L\ts1\t-\ts1\t+\t0M"]) + "\n",
    # L1 / L2 tags giving lengths to sequence-less segments; overlap longer than the segment (clamped, gfa-base.c:212-230); CRLF line ends; lines to be skipped
    "lengths_crlf": "\r\n".join([
        "H\tVN:Z:1.0", "S\tx\t*", "S\ty\t*\tLN:i:7", "S\tz\tACG", "# comment", "",
        "L\tx\t+\ty\t+\t2:3\tL1:i:8\tL2:i:4", "L\ty\t+\tz\t+\t9M", "L\tz\t+\tx\t-\t1M\tSR:i:5\tL2:i:20", "P\tpath\tx+,y+\t*"]) + "\r\n",
    # FASTA input (gfa-io.c:266-288,311-317): one segment per record, names from the running count, stable name = the record's first word; then GFA lines after it
    "fasta_then_gfa": ">chrA some description\nACGTACGTAC\nGTACGTAC\n>chrB\nTTTTGGGG\nS\tq\tCCCCAAAA\nL\ts1\t+\ts2\t+\t0M\nL\ts2\t+\tq\t-\t0M\n",
}


@pytest.mark.parametrize("name", sorted(HAND))
def test_hand_written_graphs_load_like_the_reference(name, tmp_path):
    ours, ref = libs()
    p = tmp_path / (name + ".gfa")
    p.write_bytes(HAND[name].encode())
    same_graph(snapshot(ours, str(p)), snapshot(ref, str(p)), name)


def test_reference_fixture_and_its_gzip_copy(tmp_path):
    ours, ref = libs()
    mt = os.path.join(GOLD, "MT.gfa")
    want = snapshot(ref, mt)
    same_graph(snapshot(ours, mt), want, "MT.gfa")
    gz = tmp_path / "MT.gfa.gz"
    import gzip
    gz.write_bytes(gzip.compress(open(mt, "rb").read()))
    same_graph(snapshot(ours, str(gz)), want, "MT.gfa.gz")
    fa = os.path.join(GOLD, "MT-human.fa")
    same_graph(snapshot(ours, fa), snapshot(ref, fa), "MT-human.fa (a FASTA file as the graph)")


def test_bubble_graphs_of_the_bench_generator(tmp_path):
    """the synthetic linear + bubble graphs every e2e test and the bench map against: thousands of arcs whose per-vertex order comes out of klib's unstable radix sort"""
    ours, ref = libs()
    import minigraph_amd as mga
    for k, (genome, hap, chrs) in enumerate(((400000, 3, 1), (900000, 5, 4))):
        pre = str(tmp_path / ("g%d" % k))
        subprocess.run([mga.MGSIM, "-p", pre, "-G", str(genome), "-c", str(chrs), "-H", str(hap), "-n", "1", "-s", str(7 + k)], stderr=subprocess.DEVNULL, check=True)
        same_graph(snapshot(ours, pre + ".gfa"), snapshot(ref, pre + ".gfa"), "bubble graph %d" % k)
        same_graph(snapshot(ours, pre + ".lin.fa"), snapshot(ref, pre + ".lin.fa"), "linear FASTA %d" % k)


def test_random_link_soup(tmp_path):
    """random multigraphs with random overlap spellings: duplicates, one-sided overlaps that do or do not agree with the complement line, missing complements"""
    ours, ref = libs()
    rng = np.random.default_rng(5)
    for t in range(12):
        n = int(rng.integers(3, 40))
        lines = []
        for i in range(n):
            ln = int(rng.integers(5, 60))
            seq = "".join("ACGT"[x] for x in rng.integers(0, 4, ln)) if rng.random() < 0.8 else "*\tLN:i:%d" % ln
            lines.append("S\tn%d\t%s" % (i, seq))
        for _ in range(int(rng.integers(n, 4 * n))):
            a, b = int(rng.integers(0, n + (1 if rng.random() < 0.05 else 0))), int(rng.integers(0, n))
            k = rng.random()
            ov = "0M" if k < 0.4 else "%dM" % rng.integers(0, 5) if k < 0.6 else "%d:%d" % (rng.integers(0, 4), rng.integers(0, 4)) if k < 0.75 else "%d:" % rng.integers(0, 4) if k < 0.85 else ":%d" % rng.integers(0, 4) if k < 0.95 else "*"
            tag = "\tSR:i:%d" % rng.integers(0, 4) if rng.random() < 0.5 else ""
            lines.append("L\tn%d\t%s\tn%d\t%s\t%s%s" % (a, "+-"[int(rng.integers(0, 2))], b, "+-"[int(rng.integers(0, 2))], ov, tag))
            if rng.random() < 0.3:   # its complement, sometimes with the overlaps the other way round as it should be, sometimes not
                lines.append("L\tn%d\t%s\tn%d\t%s\t%s%s" % (b, "-+"["+-".index(lines[-1].split("\t")[4])], a, "-+"["+-".index(lines[-1].split("\t")[2])], ov if rng.random() < 0.5 else "0M", tag))
        p = tmp_path / ("soup%d.gfa" % t)
        p.write_text("\n".join(lines) + "\n")
        same_graph(snapshot(ours, str(p)), snapshot(ref, str(p)), "soup %d" % t)
