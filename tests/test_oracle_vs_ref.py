"""Pin the plain-C restatement (oracle/mgo_*.c) against the UNMODIFIED reference (oracle/_ref/libmgref.so).

CPU only.  Skipped where the reference library has not been built (it needs /root/reference)."""
import numpy as np
import pytest

import refbind as rb

pytestmark = pytest.mark.skipif(not (rb.have_ref() and rb.have_oracle()), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def ref():
    return rb.Ref()


@pytest.fixture(scope="module")
def ora():
    return rb.Oracle()


def rand_seq(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n).tobytes())


def mutate(rng, s, err):
    out = bytearray()
    for ch in s:
        u = rng.random()
        if u < err * 0.4:
            out.append(rng.choice([c for c in b"ACGT" if c != ch]))
        elif u < err * 0.7:
            out.append(rng.choice(list(b"ACGT")))
            out.append(ch)
        elif u < err:
            pass
        else:
            out.append(ch)
    return bytes(out)


def test_sort128x_permutation(ref, ora):
    rng = np.random.default_rng(1)
    for n in [0, 1, 2, 63, 64, 65, 66, 200, 1000, 5000]:
        for key_bits in [3, 9, 17, 40, 64]:
            a = np.zeros(n, dtype=rb.m128)
            a["x"] = rng.integers(0, 2 ** min(key_bits, 63), size=n, dtype=np.uint64)
            if key_bits == 64:
                a["x"] |= rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63)
            a["y"] = np.arange(n, dtype=np.uint64)  # payload exposes the tie order
            r, o = ref.sort128x(a), ora.sort128x(a)
            assert np.array_equal(r, o), (n, key_bits)


def test_sort64(ref, ora):
    rng = np.random.default_rng(2)
    for n in [0, 1, 64, 65, 300, 4000]:
        a = rng.integers(0, 2 ** 40, size=n, dtype=np.uint64)
        assert np.array_equal(ref.sort64(a), ora.sort64(a))


@pytest.mark.parametrize("w,k", [(11, 17), (10, 19), (10, 21), (5, 4), (3, 6), (1, 5), (16, 28), (200, 15)])
def test_sketch_random(ref, ora, w, k):
    rng = np.random.default_rng(100 + w * 31 + k)
    for n in [1, 2, k - 1, k, k + w - 2, k + w - 1, k + w, 50, 300, 2000, 10000]:
        if n <= 0:
            continue
        for alphabet in [b"ACGT", b"ACGTN", b"AC", b"A", b"ACGTacgtNnUuRY"]:
            s = rand_seq(rng, n, alphabet)
            r, o = ref.sketch(s, w, k, 7), ora.sketch(s, w, k, 7)
            assert np.array_equal(r, o), (w, k, n, alphabet)


def test_sketch_low_complexity(ref, ora):
    rng = np.random.default_rng(5)
    unit = [b"A", b"AT", b"ACG", b"AACCGGTT", b"ACGTACGTAC"]
    for u in unit:
        for k, w in [(17, 11), (6, 4), (8, 5), (4, 3)]:
            s = (u * 400)[:1500]
            s = s[:700] + rand_seq(rng, 30) + s[700:]
            assert np.array_equal(ref.sketch(s, w, k), ora.sketch(s, w, k)), (u, k, w)


def test_wfa_random(ref, ora):
    rng = np.random.default_rng(7)
    for it in range(400):
        tl = int(rng.integers(1, 200))
        t = rand_seq(rng, tl)
        q = mutate(rng, t, float(rng.choice([0.0, 0.05, 0.1, 0.2, 0.4])))
        if len(q) == 0:
            q = b"A"
        if it % 7 == 0:
            q = rand_seq(rng, int(rng.integers(1, 200)))
        if it % 11 == 0:
            t = t[:tl // 2] + rand_seq(rng, int(rng.integers(20, 120))) + t[tl // 2:]
        rs, rc = ref.wfa(t, q)
        os_, oc = ora.wfa(t, q)
        assert rs == os_ and np.array_equal(rc, oc), (it, t, q)


def test_wfa_long_gap_and_trim(ref, ora):
    """scores >= 256 exercise the periodic band trimming (miniwfa.c:420)."""
    rng = np.random.default_rng(8)
    for it in range(12):
        t = rand_seq(rng, int(rng.integers(300, 900)))
        q = mutate(rng, t, 0.25)
        if it % 3 == 0:
            q = q[:100] + q[400:]
        if it % 4 == 0:
            q = rand_seq(rng, 300)
        if it % 5 == 0:
            t = t.replace(b"A", b"N", 3)
        rs, rc = ref.wfa(t, q)
        os_, oc = ora.wfa(t, q)
        assert rs == os_ and np.array_equal(rc, oc), it
        assert rs >= 0


def make_anchors(rng, n, n_chain=3, span=17, noise=0.3, tie_frac=0.0):
    """x-sorted anchors: a few colinear runs plus noise, as collect_seed_hits would hand to the DP."""
    xs, ys = [], []
    for c in range(n_chain):
        r0, q0 = int(rng.integers(0, 200000)), int(rng.integers(0, 3000))
        rev = int(rng.integers(0, 2))
        m = n // n_chain
        dr = rng.integers(1, 60, size=m).cumsum()
        dq = dr + rng.integers(-3, 4, size=m) * (rng.random(m) < 0.3)
        for i in range(m):
            xs.append((rev << 32) | (r0 + int(dr[i])))
            ys.append((span << 32) | max(span, q0 + int(dq[i])))
    k = int(n * noise)
    for i in range(k):
        xs.append((int(rng.integers(0, 2)) << 32) | int(rng.integers(0, 200000)))
        ys.append((span << 32) | int(rng.integers(span, 10000)))
    a = np.zeros(len(xs), dtype=rb.m128)
    a["x"], a["y"] = np.array(xs, dtype=np.uint64), np.array(ys, dtype=np.uint64)
    if tie_frac > 0:
        idx = rng.integers(0, len(a), size=int(len(a) * tie_frac))
        a["x"][idx] = a["x"][(idx + 1) % len(a)]
    return a


def test_lchain_dp(ref, ora):
    rng = np.random.default_rng(9)
    for it in range(60):
        n = int(rng.choice([5, 20, 64, 65, 130, 400, 1500]))
        a = make_anchors(rng, n, n_chain=int(rng.integers(1, 5)), tie_frac=0.05 if it % 2 else 0.0)
        a = ref.sort128x(a)
        kw = dict(max_skip=int(rng.choice([25, 2])), max_iter=int(rng.choice([5000, 20])), bw=int(rng.choice([500, 100])))
        ru, ra = ref.lchain_dp(a, **kw)
        ou, oa = ora.lchain_dp(a, **kw)
        assert np.array_equal(ru, ou), it
        assert np.array_equal(ra, oa), it


# ---- seeds: oracle/mgo_seed.c against the reference's mg_idx_get (index.c:50-72) and collect_seed_hits (map-algo.c:152-192) ----

def _capture_stderr(fn):
    """run fn() with fd 2 redirected to a file; returns what was written (the reference prints its MG_DBG_SEED dump with fprintf(stderr))"""
    import os
    import tempfile
    tmp = tempfile.TemporaryFile()
    saved = os.dup(2)
    try:
        os.dup2(tmp.fileno(), 2)
        fn()
    finally:
        os.dup2(saved, 2)
        os.close(saved)
    tmp.seek(0)
    return tmp.read().decode()


def _seed_case(ref, ora, path, segs, reads, preset=b"lr", occ_max1=None):
    import ctypes as C
    import minigraph_amd as mga
    L = ref.lib
    L.mg_opt_set.argtypes = [C.c_char_p, C.POINTER(mga.idxopt_t), C.POINTER(mga.mapopt_t), C.POINTER(mga.ggopt_t)]
    L.mg_index.argtypes = [C.c_void_p, C.POINTER(mga.idxopt_t), C.c_int, C.POINTER(mga.mapopt_t)]
    L.mg_index.restype = C.c_void_p
    L.mg_idx_destroy.argtypes = [C.c_void_p]
    L.mg_tbuf_init.restype = C.c_void_p
    L.mg_tbuf_destroy.argtypes = [C.c_void_p]
    L.mg_map.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.POINTER(mga.mapopt_t), C.c_char_p]
    L.mg_map.restype = C.c_void_p
    L.mg_gchain_free.argtypes = [C.c_void_p]
    io, mo, go = mga.idxopt_t(), mga.mapopt_t(), mga.ggopt_t()
    L.mg_opt_set(None, C.byref(io), C.byref(mo), C.byref(go))
    assert L.mg_opt_set(preset, C.byref(io), C.byref(mo), C.byref(go)) == 0
    g = L.gfa_read(path.encode())
    assert g
    gi = L.mg_index(g, C.byref(io), 2, C.byref(mo))          # applies mg_opt_update: occ_max1 from the index (index.c:74-93)
    assert gi
    if occ_max1 is not None:
        mo.occ_max1 = occ_max1
    names = [s[0] for s in segs]
    oidx = ora.idx_build([s[1] for s in segs], io.w, io.k)
    dbg = C.c_int.in_dll(L, "mg_dbg_flag")
    tb = L.mg_tbuf_init()
    n_checked = n_dropped = 0
    try:
        for qi, q in enumerate(reads):
            mz = ref.sketch(q, io.w, io.k, 0)
            # (1) mg_idx_get for every minimizer of the read: same count, same position list
            for x in mz["x"]:
                n_r, n_o = C.c_int(0), C.c_int32(0)
                pr = L.mg_idx_get(gi, int(x) >> 8, C.byref(n_r))
                po = ora.lib.mgo_idx_get(oidx, int(x) >> 8, C.byref(n_o))
                assert n_r.value == n_o.value, (qi, hex(int(x)), n_r.value, n_o.value)
                if n_r.value:
                    assert [pr[i] for i in range(n_r.value)] == [po[i] for i in range(n_o.value)]
                n_dropped += n_r.value >= mo.occ_max1
            # (2) collect_seed_hits through the reference's own MG_DBG_SEED dump (map-algo.c:370-375)
            dbg.value = 0x4
            txt = _capture_stderr(lambda: L.mg_gchain_free(L.mg_map(gi, len(q), q, tb, C.byref(mo), b"q%d" % qi)))
            dbg.value = 0
            rs = [l.split("\t") for l in txt.splitlines() if l.startswith(("RS\t", "SD\t"))]
            assert rs and rs[0][0] == "RS"
            a, rep_len, _ = ora.seed_hits(oidx, mz, mo.occ_max1)
            assert int(rs[0][1]) == rep_len, (qi, rs[0], rep_len)
            sd = rs[1:]
            assert len(sd) == len(a), (qi, len(sd), len(a))
            ax, ay = a["x"], a["y"]
            for i, f in enumerate(sd):
                got = (names[int(ax[i]) >> 33], int(np.int32(int(ax[i]) & 0xffffffff)), "+-"[int(ax[i]) >> 32 & 1],
                       int(np.int32(int(ay[i]) & 0xffffffff)), int(ay[i]) >> 32 & 0xff)
                want = (f[1], int(f[2]), f[3], int(f[4]), int(f[5]))
                assert got == want, (qi, i, got, want)
            n_checked += len(a)
    finally:
        L.mg_tbuf_destroy(tb)
        L.mg_idx_destroy(gi)
        L.gfa_destroy(g)
        ora.idx_free(oidx)
    return n_checked, n_dropped


def test_seed_hits(ref, ora, tmp_path):
    """random 3-haplotype bubble graph (mgsim) + a repeat-rich FASTA target where minimizers pass occ_max1 (rep_len, tandem flags)"""
    import subprocess
    import minigraph_amd as mga
    rng = np.random.default_rng(31)
    pre = str(tmp_path / "s")
    subprocess.check_call([mga.MGSIM, "-p", pre, "-G", "400000", "-H", "3", "-n", "12", "-l", "4000", "-s", "9"], stderr=subprocess.DEVNULL)
    segs = [(l.split("\t")[1], l.split("\t")[2].encode()) for l in open(pre + ".gfa") if l.startswith("S\t")]
    reads = [l.strip().encode() for l in open(pre + ".reads.fa") if not l.startswith(">")]
    n, _ = _seed_case(ref, ora, pre + ".gfa", segs, reads)
    assert n > 1000
    # repeats: a 700 bp unit copied 80x (with 1 % divergence) inside random flanks, on two target sequences
    unit = rand_seq(rng, 700)
    t1 = rand_seq(rng, 20000) + b"".join(mutate(rng, unit, 0.01) for _ in range(80)) + rand_seq(rng, 20000)
    t2 = rand_seq(rng, 5000) + b"".join(mutate(rng, unit, 0.02) for _ in range(30)) + rand_seq(rng, 5000)
    fa = str(tmp_path / "rep.fa")
    open(fa, "wb").write(b">t1\n" + t1 + b"\n>t2\n" + t2 + b"\n")
    reads = [mutate(rng, t1[15000:26000], 0.05), mutate(rng, t1[30000:36000], 0.08), mutate(rng, t2[2000:12000], 0.03), unit * 3]
    for occ in (None, 20, 200):
        n, dropped = _seed_case(ref, ora, fa, [("s1", t1), ("s2", t2)], reads, occ_max1=occ)   # gfa_read names FASTA records s1, s2, ... (gfa-io.c:311-322)
        assert n > 100
        if occ == 20:
            assert dropped > 100   # the occ filter (map-algo.c:72-79) was exercised
