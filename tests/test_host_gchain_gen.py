"""CPU: the assembly half of graph chaining (gc_core.h: gc_assemble_begin / gc_assemble_run) against the reference's own
mg_gchain_gen (gchain1.c:443-520) on hand-made chain records whose JUNCTIONS OVERLAP: the anchor count that decides whether a
group becomes a graph chain and the record hash are both taken BEFORE resolve_overlap trims the junction (gchain1.c:452-456,
472-484) -- a hash taken afterwards changes the (score, hash) order of equal-score chains, a count taken afterwards drops a chain
the reference keeps."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

import minigraph_amd as mga
import refbind as rb


class lchain_t(C.Structure):  # mg_lchain_t, minigraph.h:100-106
    _fields_ = [("off", C.c_int32), ("cnt", C.c_int32, 31), ("inner_pre", C.c_int32, 1), ("v", C.c_uint32),
                ("rs", C.c_int32), ("re", C.c_int32), ("qs", C.c_int32), ("qe", C.c_int32),
                ("score", C.c_int32), ("dist_pre", C.c_int32), ("hash_pre", C.c_uint32)]


class llchain_t(C.Structure):
    _fields_ = [("off", C.c_int32), ("cnt", C.c_int32), ("v", C.c_uint32), ("score", C.c_int32), ("ed", C.c_int32)]


class gchain_t(C.Structure):  # mg_gchain_t, minigraph.h:126-139
    _fields_ = [("id", C.c_int32), ("parent", C.c_int32), ("off", C.c_int32), ("cnt", C.c_int32), ("n_anchor", C.c_int32), ("score", C.c_int32),
                ("qs", C.c_int32), ("qe", C.c_int32), ("plen", C.c_int32), ("ps", C.c_int32), ("pe", C.c_int32), ("blen", C.c_int32), ("mlen", C.c_int32),
                ("div", C.c_float), ("hash", C.c_uint32), ("subsc", C.c_int32), ("n_sub", C.c_int32), ("bits", C.c_uint32),
                ("p", C.c_void_p), ("ds_len", C.c_int32), ("ds_n_off", C.c_int32), ("ds_off", C.c_void_p), ("ds_ds", C.c_void_p)]


class gchains_t(C.Structure):
    _fields_ = [("km", C.c_void_p), ("n_gc", C.c_int32), ("n_lc", C.c_int32), ("n_a", C.c_int32), ("rep_len", C.c_int32),
                ("gc", C.POINTER(gchain_t)), ("lc", C.POINTER(llchain_t)), ("a", C.c_void_p)]


SPAN = 17


def anchors(pts, rank0=0):
    """(rpos, qpos) END positions -> mg128_t[] as mg_update_anchors leaves them: x = minimizer rank << 32 | rpos, y = span << 32 | qpos"""
    a = np.zeros(len(pts), dtype=rb.m128)
    for i, (r, q) in enumerate(pts):
        a[i]["x"] = (rank0 + i) << 32 | r
        a[i]["y"] = SPAN << 32 | q
    return a


def chain_rec(a, off, cnt, v, score):
    c = lchain_t()
    c.off, c.cnt, c.inner_pre, c.v, c.score, c.dist_pre, c.hash_pre = off, cnt, 0, v, score, -1, 0
    c.rs = int(a[off]["x"] & 0xffffffff) + 1 - SPAN
    c.qs = int(a[off]["y"] & 0xffffffff) + 1 - SPAN
    c.re = int(a[off + cnt - 1]["x"] & 0xffffffff) + 1
    c.qe = int(a[off + cnt - 1]["y"] & 0xffffffff) + 1
    return c


def snapshot(gs):
    out = []
    for i in range(gs.n_gc):
        g = gs.gc[i]
        lcs = [(gs.lc[g.off + j].off, gs.lc[g.off + j].cnt, gs.lc[g.off + j].v, gs.lc[g.off + j].score, gs.lc[g.off + j].ed) for j in range(g.cnt)]
        out.append((g.off, g.cnt, g.n_anchor, g.score, g.qs, g.qe, g.plen, g.ps, g.pe, g.blen, g.mlen, g.hash, round(g.div, 6), tuple(lcs)))
    a = np.ctypeslib.as_array(C.cast(gs.a, C.POINTER(C.c_uint64)), shape=(gs.n_a * 2,)).copy() if gs.n_a > 0 else np.zeros(0, np.uint64)
    return out, a


def run_both(n_u_groups, chains, a, hash_, min_gc_cnt, min_gc_score, qlen=2000):
    L, R = mga.load(), rb.Ref().lib
    d = tempfile.mkdtemp()
    gfa = os.path.join(d, "g.gfa")
    rng = np.random.default_rng(5)
    with open(gfa, "w") as f:
        for s in range(2):
            f.write("S\ts%d\t%s\n" % (s, "".join(rng.choice(list("ACGT"), 3000))))
        f.write("L\ts0\t+\ts1\t+\t0M\n")
    qseq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), qlen).tobytes())
    u = (C.c_uint64 * len(n_u_groups))(*[(sc << 32) | n for sc, n in n_u_groups])
    lc_t = lchain_t * len(chains)
    # ours
    io, mo, go = mga.idxopt_t(), mga.mapopt_t(), mga.ggopt_t()
    L.mg_opt_set(None, C.byref(io), C.byref(mo), C.byref(go))
    L.gfa_read.restype = C.c_void_p
    L.mga_idx_hostpart.restype = C.c_void_p
    L.mga_idx_hostpart.argtypes = [C.c_void_p, C.c_void_p]
    g = L.gfa_read(gfa.encode())
    gi = L.mga_idx_hostpart(g, C.byref(io))
    es = C.cast(gi, C.POINTER(C.c_void_p))[1]
    L.mga_gchain_gen_host.restype = C.POINTER(gchains_t)
    L.mga_gchain_gen_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_char_p, C.POINTER(C.c_int32)]
    lc1 = lc_t(*chains)
    a1 = a.copy()
    rc = C.c_int32(-1)
    ours = L.mga_gchain_gen_host(g, es, len(n_u_groups), u, lc1, a1.ctypes.data, len(a1), hash_, min_gc_cnt, min_gc_score, 10000, qseq, C.byref(rc))
    assert rc.value == 0
    # reference (it modifies lc[] in place: its own copy)
    R.gfa_read.restype = C.c_void_p
    R.gfa_edseq_init.restype = C.c_void_p
    R.gfa_edseq_init.argtypes = [C.c_void_p]
    R.mg_gchain_gen.restype = C.POINTER(gchains_t)
    R.mg_gchain_gen.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int32, C.c_char_p]
    gr = R.gfa_read(gfa.encode())
    esr = R.gfa_edseq_init(gr)
    lc2 = lc_t(*chains)
    a2 = a.copy()
    ref = R.mg_gchain_gen(None, None, gr, esr, len(n_u_groups), u, lc2, a2.ctypes.data, hash_, min_gc_cnt, min_gc_score, 10000, 1, qseq)
    so, ao = snapshot(ours.contents)
    sr, ar = snapshot(ref.contents)
    return so, ao, sr, ar


@pytest.mark.skipif(not rb.have_ref(), reason="oracle/_ref not built")
def test_overlapping_junction_count_and_hash_are_taken_before_the_trim():
    # chain 0: six anchors up to (250, 250); chain 1 starts at (200, 200): the junction trims two anchors off each side (11 -> 7 anchors)
    pts0 = [(100 + 30 * i, 100 + 30 * i) for i in range(6)]
    pts1 = [(200 + 30 * i, 200 + 30 * i) for i in range(5)]
    a = anchors(pts0 + pts1)
    chains = [chain_rec(a, 0, 6, 0, 100), chain_rec(a, 6, 5, 0, 80)]
    # min_gc_cnt 9: kept by its 11 anchors although only 7 survive the trim
    so, ao, sr, ar = run_both([(150, 2)], chains, a, 0x1234567, 9, 50)
    assert len(sr) == 1 and sr[0][2] < 9, "the case must trim below min_gc_cnt in the reference"
    assert so == sr and np.array_equal(ao, ar)


@pytest.mark.skipif(not rb.have_ref(), reason="oracle/_ref not built")
def test_equal_scores_are_ordered_by_the_untrimmed_hash():
    # several groups of EQUAL score, each with an overlapping junction: their order is the order of the hashes of the untrimmed records
    pts, chains, groups = [], [], []
    for gidx in range(6):
        base = 40 + 330 * gidx
        p0 = [(base + 30 * i, base + 30 * i) for i in range(6)]
        p1 = [(base + 100 + 30 * i + gidx, base + 100 + 30 * i + gidx) for i in range(5)]
        pts += p0 + p1
    a = anchors(pts)
    for gidx in range(6):
        chains += [chain_rec(a, 11 * gidx, 6, 0, 100), chain_rec(a, 11 * gidx + 6, 5, 0, 80)]
        groups.append((150, 2))
    so, ao, sr, ar = run_both(groups, chains, a, 99, 5, 50)
    assert len(sr) == 6
    assert [x[11] for x in so] == [x[11] for x in sr], "hashes / order"
    assert so == sr and np.array_equal(ao, ar)


@pytest.mark.skipif(not rb.have_ref(), reason="oracle/_ref not built")
def test_groups_below_the_thresholds_are_skipped_alike():
    pts0 = [(100 + 30 * i, 100 + 30 * i) for i in range(6)]
    pts1 = [(700 + 30 * i, 700 + 30 * i) for i in range(3)]
    a = anchors(pts0 + pts1)
    chains = [chain_rec(a, 0, 6, 0, 100), chain_rec(a, 6, 3, 0, 45)]
    so, ao, sr, ar = run_both([(100, 1), (45, 1)], chains, a, 7, 5, 50)
    assert len(sr) == 1
    assert so == sr and np.array_equal(ao, ar)
    so, ao, sr, ar = run_both([(100, 1), (45, 1)], chains, a, 7, 7, 50)   # nothing passes
    assert len(sr) == 0 and so == sr
