"""GPU parity tests: every HIP stage kernel, called through the C ABI, against the oracle restatement
(and the unmodified reference library where it travelled) on the same seeded inputs.  Bit-exact."""
import numpy as np
import pytest

import minigraph_amd as mga
import refbind as rb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ora():
    return rb.Oracle()


def rand_seq(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n).tobytes())


def mutate(rng, s, err):
    u = rng.random(len(s))
    out = bytearray()
    for ch, x in zip(s, u):
        if x < err * 0.4:
            out.append(int(rng.choice([c for c in b"ACGT" if c != ch])))
        elif x < err * 0.7:
            out.append(int(rng.choice(list(b"ACGT"))))
            out.append(ch)
        elif x < err:
            pass
        else:
            out.append(ch)
    return bytes(out)


SKETCH_FORMS = ["planes", "v1", "2bit"]


def sketch_form(monkeypatch, form):
    """round 6: k_sketch takes its k-mers from bit planes (ballots of the step's codes; the LDS code ring only around ambiguous bases and at a sequence's start) -- "planes", the
    default; "v1" is the kernel of rounds 1-5 (MGA_SKETCH_V1=1), "2bit" reads the bases themselves as packed bit planes made by k_pack2 (MGA_SKETCH_2BIT=1): same minimizers"""
    if form == "v1":
        monkeypatch.setenv("MGA_SKETCH_V1", "1")
    elif form == "2bit":
        monkeypatch.setenv("MGA_SKETCH_2BIT", "1")


@pytest.mark.parametrize("form", SKETCH_FORMS)
@pytest.mark.parametrize("w,k", [(11, 17), (10, 19), (10, 21), (5, 4), (3, 6), (1, 5), (16, 28), (200, 15), (255, 28), (63, 27), (64, 27)])
def test_sketch_parity(ora, w, k, form, monkeypatch):
    sketch_form(monkeypatch, form)
    rng = np.random.default_rng(1000 + w * 31 + k)
    seqs = []
    for gap in (40, 70, 200, 700):   # sparse ambiguous bases: the steps switch between the plane form and the code ring, with k - 1 .. 64 real bases in between
        s = bytearray(rand_seq(rng, 4000))
        pos = int(rng.integers(0, gap))
        while pos < len(s):
            s[pos:pos + int(rng.integers(1, 3))] = b"N"
            pos += int(rng.integers(max(1, gap // 2), gap * 2))
        seqs.append(bytes(s))
        seqs.append(bytes(s[:64 + k]) + b"N" + bytes(s[:200]))
    for n in [1, 2, k - 1, k, k + w - 2, k + w - 1, k + w, 63, 64, 65, 127, 128, 129, 300, 2000, 10000]:
        if n > 0:
            for alphabet in [b"ACGT", b"ACGTN", b"AC", b"A", b"ACGTacgtNnUuRY"]:
                seqs.append(rand_seq(rng, n, alphabet))
    for u in [b"A", b"AT", b"ACG", b"AACCGGTT", b"ACGTACGTAC"]:
        s = (u * 400)[:1500]
        seqs.append(s[:700] + rand_seq(rng, 30) + s[700:])
    got = mga.sketch_batch(seqs, w, k, rid=np.arange(len(seqs)) % 7)
    for i, s in enumerate(seqs):
        exp = ora.sketch(s, w, k, i % 7)
        assert np.array_equal(got[i], exp), (w, k, len(s), s[:40])


@pytest.mark.parametrize("form", SKETCH_FORMS)
def test_sketch_many_reads(ora, form, monkeypatch):
    sketch_form(monkeypatch, form)
    rng = np.random.default_rng(5)
    seqs = [rand_seq(rng, int(rng.integers(9000, 11000))) for _ in range(300)]
    got = mga.sketch_batch(seqs, 11, 17)
    for s, g in zip(seqs, got):
        assert np.array_equal(g, ora.sketch(s, 11, 17))


def wfa_cases(rng, n, maxlen):
    T, Q = [], []
    for it in range(n):
        tl = int(rng.integers(1, maxlen))
        t = rand_seq(rng, tl)
        q = mutate(rng, t, float(rng.choice([0.0, 0.05, 0.1, 0.2, 0.4])))
        if len(q) == 0:
            q = b"A"
        if it % 7 == 0:
            q = rand_seq(rng, int(rng.integers(1, maxlen)))
        if it % 11 == 0:
            t = t[:tl // 2] + rand_seq(rng, int(rng.integers(20, 120))) + t[tl // 2:]
        if it % 13 == 0:
            t = t.replace(b"A", b"N", 2)
            q = q.replace(b"C", b"N", 1)
        T.append(t)
        Q.append(q)
    return T, Q


def test_wfa_parity_small(ora):
    rng = np.random.default_rng(7)
    T, Q = wfa_cases(rng, 1500, 200)
    sc, cg = mga.wfa_batch(T, Q)
    for i in range(len(T)):
        es, ec = ora.wfa(T[i], Q[i])
        assert es == sc[i], (i, T[i], Q[i])
        assert np.array_equal(ec, cg[i]), (i, T[i], Q[i])


def test_wfa_parity_tiers(ora):
    """bands > 256 diagonals / scores >= 256 (trimming) leave tier 0 and must still be exact"""
    rng = np.random.default_rng(8)
    T, Q = [], []
    for it in range(24):
        t = rand_seq(rng, int(rng.integers(300, 1500)))
        q = mutate(rng, t, 0.25)
        if it % 3 == 0:
            q = q[:100] + q[400:]
        if it % 4 == 0:
            q = rand_seq(rng, 300)
        if it % 5 == 0:
            t = t.replace(b"A", b"N", 3)
        T.append(t)
        Q.append(q)
    # one long, clean pair: long match runs, tiny score
    t = rand_seq(rng, 20000)
    T.append(t)
    Q.append(t[:9000] + b"G" + t[9000:15000] + t[15010:])
    sc, cg = mga.wfa_batch(T, Q)
    for i in range(len(T)):
        es, ec = ora.wfa(T[i], Q[i])
        assert es == sc[i], i
        assert np.array_equal(ec, cg[i]), i


def test_wfa_roundtrip_property():
    """size-independent property: every CIGAR consumes exactly tl/ql bases and re-scores to the reported penalty"""
    rng = np.random.default_rng(9)
    T, Q = wfa_cases(rng, 5000, 150)
    sc, cg = mga.wfa_batch(T, Q)
    for i in range(len(T)):
        t, q, ti, qi, pen = T[i], Q[i], 0, 0, 0
        for op in cg[i]:
            l, o = int(op) >> 4, int(op) & 15
            if o == 7:
                assert t[ti:ti + l] == q[qi:qi + l]
                ti += l; qi += l
            elif o == 8:
                assert all(t[ti + j] != q[qi + j] for j in range(l))
                ti += l; qi += l; pen += 4 * l
            elif o == 1:
                qi += l; pen += min(4 + 2 * l, 15 + l)
            elif o == 2:
                ti += l; pen += min(4 + 2 * l, 15 + l)
            else:
                raise AssertionError(o)
        assert ti == len(t) and qi == len(q) and pen == sc[i], i


@pytest.mark.parametrize("packed", ["0", "5", None, "31", "7+own_walk"])
def test_wfa_windowed_tiers_edge_shapes(ora, monkeypatch, packed):
    """(round 6, "7+own_walk" = MGA_WFA_FUSE_TB=1: the packed rungs of 128 / 192 / 256 diagonals walk their own alignments behind the forward pass instead of leaving them to
    k_wfa_tb -- measured slower, not the default, same results)
    (MGA_WFA_PACKED: every rung on the one-diagonal-per-lane kernel, two of the wide rungs packed, the default, and everything that has a packed form -- the rungs of
    32 and 64 diagonals with four / two problems per wavefront included)
    the windowed tiers (k_wfa_w.hip: 16 / 32 / 64 / 128 / 192 / 256 diagonals, several problems per wavefront in the narrow ones) are exact only
    below the bound of their window: single gaps of every length around each half-width (the alignment hugs the window's edge, one base further and
    the problem must give up and climb), the same with noise, matrices narrower than the window, end diagonals far from 0 (the window is centred
    between 0 and ql - tl), sequences longer than a tier's LDS staging, scores around 256 (the last windowed score), N bases"""
    if packed is None:
        monkeypatch.delenv("MGA_WFA_PACKED", raising=False)
    elif packed == "7+own_walk":
        monkeypatch.setenv("MGA_WFA_PACKED", "7")
        monkeypatch.setenv("MGA_WFA_FUSE_TB", "1")
    else:
        monkeypatch.setenv("MGA_WFA_PACKED", packed)
    rng = np.random.default_rng(41)
    T, Q = [], []
    for L in (12, 30, 60, 70, 72, 100, 111, 112, 128, 129, 167, 168, 192, 193, 255, 256, 257, 343, 344, 384, 385, 512, 513, 700):
        t = rand_seq(rng, L)
        for g in (1, 2, 6, 7, 8, 9, 10, 14, 15, 16, 17, 18, 30, 31, 32, 33, 34, 62, 63, 64, 65, 66, 94, 95, 96, 97, 98, 126, 127, 128, 129, 130, 160, 190, 222, 250):
            if g >= L:
                continue
            cut = int(rng.integers(0, L - g + 1))
            T.append(t); Q.append(t[:cut] + t[cut + g:])
            T.append(t[:cut] + t[cut + g:]); Q.append(t)
            m = mutate(rng, t, 0.08)
            c2 = min(cut, max(0, len(m) - g))
            T.append(t); Q.append(m[:c2] + m[c2 + g:] if len(m) > g else m)
        for short in (1, 2, 5, 15, 16, 17, 31, 33):
            T.append(rand_seq(rng, short)); Q.append(t)
            T.append(t); Q.append(rand_seq(rng, short))
            T.append(rand_seq(rng, short)); Q.append(rand_seq(rng, short))
        for err in (0.02, 0.1, 0.2, 0.3, 0.45):
            T.append(t); Q.append(mutate(rng, t, err) or b"A")
        T.append(t.replace(b"A", b"N", 3)); Q.append(mutate(rng, t, 0.1).replace(b"C", b"N", 2) or b"N")
    sc, cg = mga.wfa_batch(T, Q)
    for i in range(len(T)):
        es, ec = ora.wfa(T[i], Q[i])
        assert es == sc[i], (i, len(T[i]), len(Q[i]), es, sc[i])
        assert np.array_equal(ec, cg[i]), (i, len(T[i]), len(Q[i]), es)


@pytest.mark.parametrize("packed", ["0", "7", "31"])
def test_wfa_windowed_tiers_many_problems(ora, monkeypatch, packed):
    """a launch the size of a small chunk with the bench workload's shape (gap lengths 1..400, 10 % errors): every group of every wavefront is refilled
    many times, the queue runs dry at the end, problems climb from tier to tier"""
    monkeypatch.setenv("MGA_WFA_PACKED", packed)
    rng = np.random.default_rng(43)
    T, Q = [], []
    for it in range(30000):
        L = int(min(400, 1 + rng.exponential(70)))
        t = rand_seq(rng, L)
        q = mutate(rng, t, 0.1) or b"A"
        T.append(t); Q.append(q)
    sc, cg = mga.wfa_batch(T, Q)
    bad = 0
    for i in range(len(T)):
        es, ec = ora.wfa(T[i], Q[i])
        if es != sc[i] or not np.array_equal(ec, cg[i]):
            bad += 1
            assert bad < 5, (i, len(T[i]), len(Q[i]), es, sc[i])
    assert bad == 0


def test_wfa_ladder_list_overflow():
    """a rung's work list has room for its own problems + a share of what the rungs below it run; with MGA_WFA_ARRIVALS_PCT=0 (4096 arrivals) 12 000 unrelated
    pairs overflow every list on their way up: the sweep leaves them open and the slices with room for everything finish them -- same answers"""
    import os, subprocess, sys
    env = dict(os.environ, MGA_WFA_ARRIVALS_PCT="0", MGA_DEBUG_WFA="1")
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "wfa_overflow_child.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"OVERFLOW-OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])
    assert p.stderr.count(b"[wfa] sweep over") >= 2, p.stderr[-2000:]   # the second sweep happened


# ---------------------------------------------------------------------------------------------
# seeds + linear chaining against a real (synthetic) graph
# ---------------------------------------------------------------------------------------------
import os
import subprocess
import tempfile


def read_fa(path):
    seqs, cur = [], []
    for line in open(path, "rb"):
        if line.startswith(b">"):
            if cur:
                seqs.append(b"".join(cur))
            cur = []
        else:
            cur.append(line.strip())
    if cur:
        seqs.append(b"".join(cur))
    return seqs


def read_gfa_segs(path):
    return [l.split(b"\t")[2] for l in open(path, "rb") if l.startswith(b"S\t")]


@pytest.fixture(scope="module")
def sim():
    d = tempfile.mkdtemp(prefix="mga_sim_")
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "1500000", "-H", "3", "-n", "300", "-s", "7"],
                          stderr=subprocess.DEVNULL)
    # make some minimizers repetitive: append a tandem-ish segment family to the graph
    return d


@pytest.mark.parametrize("long_path", [False, True])
def test_seed_and_lchain_parity(ora, sim, long_path, monkeypatch):
    """long_path: the intra-read parallel seed kernels that -x asm uses for contigs of megabases (probe per minimizer, device scans,
    rep_len from a running maximum), here on ordinary reads against the same oracle"""
    if long_path:
        monkeypatch.setenv("MGA_SEED_LONG", "1")
    monkeypatch.setenv("MGA_LC_WIN", "0" if long_path else "1")   # (the first-pass DP over memory / with the last 64 anchors in registers)
    gfa, reads = os.path.join(sim, "t.gfa"), read_fa(os.path.join(sim, "t.reads.fa"))
    G = mga.Graph(gfa)
    try:
        segs = read_gfa_segs(gfa)
        oidx = ora.idx_build(segs, 11, 17)
        mz = mga.sketch_batch(reads, 11, 17)
        for max_occ in (G.mo.occ_max1, 2):
            got = G.seed_batch(mz, max_occ=max_occ)
            exp = [ora.seed_hits(oidx, m, max_occ) for m in mz]
            for i in range(len(reads)):
                assert got[i][1] == exp[i][1], ("rep_len", i)
                assert np.array_equal(got[i][2], exp[i][2]), ("mini_pos", i)
                assert np.array_equal(got[i][0], exp[i][0]), ("anchors", i)
        anchors = [e[0] for e in exp]
        got = G.seed_batch(mz)
        anchors = [g[0] for g in got]
        for kw in (dict(), dict(max_skip=2, bw=100), dict(max_iter=20)):
            lc = mga.lchain_batch(anchors, **kw)
            for i in range(len(reads)):
                eu, ea = ora.lchain_dp(anchors[i], **kw)
                assert np.array_equal(lc[i][0], eu), ("u", i, kw)
                assert np.array_equal(lc[i][1], ea), ("a", i, kw)
        ora.idx_free(oidx)
    finally:
        G.close()


def make_anchors(rng, n, n_chain=3, span=17, noise=0.3, tie_frac=0.0):
    xs, ys = [], []
    for c in range(n_chain):
        r0, q0 = int(rng.integers(0, 200000)), int(rng.integers(0, 3000))
        rev = int(rng.integers(0, 2))
        m = max(n // n_chain, 1)
        dr = rng.integers(1, 60, size=m).cumsum()
        dq = dr + rng.integers(-3, 4, size=m) * (rng.random(m) < 0.3)
        for i in range(m):
            xs.append((rev << 32) | (r0 + int(dr[i])))
            ys.append((span << 32) | max(span, q0 + int(dq[i])))
    for i in range(int(n * noise)):
        xs.append((int(rng.integers(0, 2)) << 32) | int(rng.integers(0, 200000)))
        ys.append((span << 32) | int(rng.integers(span, 10000)))
    a = np.zeros(len(xs), dtype=mga.m128)
    a["x"], a["y"] = np.array(xs, dtype=np.uint64), np.array(ys, dtype=np.uint64)
    if tie_frac > 0:
        idx = rng.integers(0, len(a), size=int(len(a) * tie_frac))
        a["x"][idx] = a["x"][(idx + 1) % len(a)]
    return a


@pytest.mark.parametrize("pair,win", [("0", "1"), ("0", "0"), ("1", "0")])
def test_lchain_synthetic_anchor_sets(ora, pair, win, monkeypatch):
    """ties in x and in score, tiny / large anchor sets, skip + iteration caps; pair = 1: the first-pass DP of two reads per wavefront in 32-lane groups (k_lchain2, round 5:
    measured slower than one read per wavefront and not the default, but exact -- the block of predecessors is 32 instead of 64, the replay carries its state across blocks);
    win = 1 (round 6, the default): the last 64 anchors in registers (lc_dp_w: first block of predecessors from the lanes, marks through an LDS ring, blocks further back and
    windows of hundreds of anchors -- these sets have them -- from memory), win = 0: every block from memory (lc_dp)"""
    monkeypatch.setenv("MGA_LC_PAIR", pair)
    monkeypatch.setenv("MGA_LC_WIN", win)
    rng = np.random.default_rng(9)
    sets = []
    for it in range(120):
        n = int(rng.choice([5, 20, 64, 65, 130, 400, 1500, 4000]))
        a = make_anchors(rng, n, n_chain=int(rng.integers(1, 5)), tie_frac=0.05 if it % 2 else 0.0)
        sets.append(ora.sort128x(a))
    sets.append(np.zeros(0, dtype=mga.m128))
    for kw in (dict(), dict(max_skip=2, bw=100), dict(max_iter=20), dict(min_cnt=2, min_sc=10)):
        lc = mga.lchain_batch(sets, **kw)
        for i, a in enumerate(sets):
            eu, ea = ora.lchain_dp(a, **kw)
            assert np.array_equal(lc[i][0], eu), ("u", i, kw)
            assert np.array_equal(lc[i][1], ea), ("a", i, kw)


@pytest.mark.gpu
def test_gaf_div_text():
    """dv:f: as the device's GAF writer prints it (k_gaf.hip: a float times 10^4 is exact in a double, rounded half-to-even) against "%.4f" of the same float -- random
    values, the decimal ties a float can hit exactly (odd multiples of 1/32 ... 1/4096: x.xxxx5 with nothing behind it), values next to them, the ends of [0, 1]"""
    rng = np.random.default_rng(5)
    vals = [0.0, 1.0, 0.5, 0.25, 1e-7, 4.9e-5, 5.1e-5, 0.99995, 0.999949, 0.99996, 1.0 - 2.0 ** -24]
    for sh in range(5, 13):
        vals += [m / 2.0 ** sh for m in range(1, 2 ** sh, 2) if m < 600]
    ties = np.array(vals, dtype=np.float32)
    near = np.concatenate([np.nextafter(ties, np.float32(0)), np.nextafter(ties, np.float32(2))])
    allv = np.concatenate([ties, near[(near >= 0) & (near <= 1)], rng.random(20000).astype(np.float32), (rng.random(5000) * 0.2).astype(np.float32)])
    got = mga.gaf_div_batch(allv)
    for v, g in zip(allv.tolist(), got):
        want = b"0" if v == 0.0 else ("%.4f" % v).encode()
        assert g == want, (v, g, want)


@pytest.mark.gpu
def test_device_klib_sort(ora):
    """the kernels' radix_sort_128x (dev_klibsort.h) against the restatement pinned to the reference's: the order it leaves EQUAL keys in is observable (chain ends of equal
    score, anchors of equal x), so y carries each element's original place and must come out where klib puts it.  Sizes around the insertion-sort limit (64), the LDS form's
    limit (1024: klib_sort128x_small, round 6) and beyond (the in-memory form); keys that differ in one byte only, in every byte, scores (small integers with many ties),
    anchors (segment in the high word), all equal"""
    rng = np.random.default_rng(17)
    arrays = []
    for n in [0, 1, 2, 63, 64, 65, 66, 127, 128, 129, 200, 255, 256, 257, 511, 640, 1000, 1023, 1024, 1025, 1500, 3000, 9000]:
        for kind in range(9):
            a = np.zeros(n, dtype=mga.m128)
            if kind == 0:
                x = rng.integers(0, 1 << 62, size=n, dtype=np.uint64) * np.uint64(4)           # every byte differs
            elif kind == 1:
                x = rng.integers(40, 400, size=n).astype(np.uint64)                            # scores: one byte, many ties
            elif kind == 2:
                x = rng.integers(40, 12000, size=n).astype(np.uint64)                          # scores: two bytes
            elif kind == 3:
                x = (rng.integers(0, 6, size=n).astype(np.uint64) << np.uint64(33)) | rng.integers(0, 50000, size=n).astype(np.uint64)  # anchors: (segment, strand) | position
            elif kind == 4:
                x = np.full(n, 0x1234567890, dtype=np.uint64)                                  # all equal
            elif kind == 5:
                x = rng.integers(0, 3, size=n).astype(np.uint64) << np.uint64(56)              # the top byte only: three buckets, everything else ties
            elif kind == 6:
                x = (rng.integers(0, 256, size=n).astype(np.uint64) << np.uint64(24)) | rng.integers(0, 2, size=n).astype(np.uint64)  # a middle byte, then a byte five levels down
            elif kind == 7:
                x = np.sort(rng.integers(0, 1 << 40, size=n).astype(np.uint64))[::-1].copy()   # descending
            else:
                x = (rng.integers(0, 2, size=n).astype(np.uint64) << np.uint64(8)) | np.where(rng.random(n) < 0.9, 7, rng.integers(0, 256, size=n)).astype(np.uint64)  # one bucket of > 64 among small ones
            a["x"], a["y"] = x, np.arange(n, dtype=np.uint64)
            arrays.append(a)
    got = mga.sort128x_batch(arrays)
    for i, a in enumerate(arrays):
        want = ora.sort128x(a.copy())
        assert np.array_equal(got[i]["x"], want["x"]) and np.array_equal(got[i]["y"], want["y"]), (i, len(a), i % 9)


@pytest.mark.parametrize("form", SKETCH_FORMS)
@pytest.mark.parametrize("w,k", [(11, 17), (10, 19), (5, 15), (200, 27)])
def test_sketch_long_sequences_in_pieces(ora, w, k, form, monkeypatch):
    sketch_form(monkeypatch, form)
    """sequences above 64 kb are sketched in pieces that warm up on the preceding w+k+64 bases (k odd); piece boundaries,
    N runs across them and low-complexity stretches must not show"""
    rng = np.random.default_rng(77 + w + k)

    def seq(n):
        s = bytearray(rand_seq(rng, n))
        for pos in (65536 - 9, 65536, 65536 + 3, 131072 - (w + k), 131072 - 1, 196608 - k, 200000):
            if pos + 40 < n:
                m = int(rng.integers(1, 40))
                s[pos:pos + m] = b"N" * m   # short ambiguous runs around the piece boundaries
        if n > 140000:
            s[131072 - 300:131072 + 300] = (b"AT" * 300)          # a repeat spanning a boundary: many equal minimizers in one window
        return bytes(s)
    seqs = [seq(n) for n in (65535, 65536, 65537, 131072 + 5, 300000, 1000, 70000)]
    got = mga.sketch_batch(seqs, w, k, rid=np.arange(len(seqs)))
    for i, s in enumerate(seqs):
        assert np.array_equal(got[i], ora.sketch(s, w, k, i)), (w, k, len(s))


def test_wfa_chained_fallback_matches_mwf_wfa_auto():
    """gaps whose exact WFA passes 1e8 cells take miniwfa's chained fallback (miniwfa.c:829-832): the plan is made on the host
    (wfachain.c), every stretch of it runs through the device ladder again (k_wfa_sched.hip: wfs_fallback), the stitched CIGAR
    must equal the reference's mwf_wfa_auto() -- one huge stretch (unbounded tier), the D+I shortcut for unrelated >= 10 kb,
    a mixed batch around them"""
    ref = rb.Ref()
    rng = np.random.default_rng(7)

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

    T, Q = wfa_cases(rng, 40, 300)
    big = []
    for n, sub, indel in [(9000, 0.3, 0.1), (12000, 0.3, 0.1), (9500, 0.75, 0.0)]:
        a, m, b = rand_seq(rng, 500), rand_seq(rng, n), rand_seq(rng, 500)
        big.append(len(T))
        T.insert(len(T), a + m + b)
        Q.insert(len(Q), a + mut(m, sub, indel) + b)
        t2, q2 = wfa_cases(rng, 10, 300)
        T += t2
        Q += q2
    sc, cg = mga.wfa_batch(T, Q)
    for i in range(len(T)):
        es, ec = ref.wfa(T[i], Q[i])
        assert es == sc[i], i
        assert np.array_equal(ec, cg[i]), i
    for i in big:  # the exact pass alone really gives up on these
        assert rb.Oracle().wfa(T[i], Q[i], max_iter=100000000)[0] < 0

@pytest.mark.gpu
def test_wfa_chained_fallback_200kb_pair_at_15_percent():
    """a 200 kb x 200 kb pair at 15 % divergence -- 4 x 10^10 cells for the plain exact pass -- equals mwf_wfa_auto(): its 1e8-cell cap sends it through
    the chained fallback (13-mer anchors, miniwfa.c:776-822), whose stretches the reference closes in its low-memory mode and the device in its ladder (same CIGARs: DESIGN 4)"""
    ref = rb.Ref()
    rng = np.random.default_rng(11)
    t = rand_seq(rng, 200000)
    q = bytearray()
    for c in t:
        r = rng.random()
        if r < 0.06:
            q.append(int(rng.choice([x for x in b"ACGT" if x != c])))
        elif r < 0.105:
            continue
        elif r < 0.15:
            q.append(c)
            q.append(int(rng.choice(list(b"ACGT"))))
        else:
            q.append(c)
    q = bytes(q)
    sc, cg = mga.wfa_batch([t], [q])
    es, ec = ref.wfa(t, q)
    assert es == sc[0] and np.array_equal(ec, cg[0])
    assert rb.Oracle().wfa(t[:60000], q[:60000], max_iter=100000000)[0] < 0   # (already a 60 kb prefix is beyond the exact pass's cap)


@pytest.mark.gpu
def test_wfa_parity_ring_layout_shapes(ora):
    """the register tiers fold their window of diagonals into rings around a centre (k_wfa_r.hip): shapes that push the band to one side (short target,
    long query and the reverse: the centre moves off diagonal 0), to the window's edge (the problem leaves the tier mid-way) and across every ring
    boundary (multiples of 32 diagonals), for every tier size"""
    rng = np.random.default_rng(31)
    T, Q = [], []
    for L in (40, 64, 100, 128, 160, 192, 250, 256, 400, 512, 900, 1024, 1800):
        t = rand_seq(rng, L)
        for g in (1, 15, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 129, 200):
            if g >= L:
                continue
            cut = int(rng.integers(0, L - g + 1))
            T.append(t); Q.append(t[:cut] + t[cut + g:])                       # deletion of g bases: band grows to -g
            T.append(t[:cut] + t[cut + g:]); Q.append(t)                       # insertion of g bases: band grows to +g
            T.append(mutate(rng, t, 0.1)[:max(1, L - g)]); Q.append(t)         # + noise, end clipped
        for short in (1, 3, 17, 40):
            T.append(rand_seq(rng, short)); Q.append(t)                        # matrix narrower than the window on the target side
            T.append(t); Q.append(rand_seq(rng, short))                        # ... on the query side
        T.append(t); Q.append(rand_seq(rng, L))                                # unrelated: the band runs to the window's edge
        T.append(t); Q.append(mutate(rng, t, 0.3))
    sc, cg = mga.wfa_batch(T, Q)
    for i in range(len(T)):
        es, ec = ora.wfa(T[i], Q[i])
        assert es == sc[i], (i, len(T[i]), len(Q[i]))
        assert np.array_equal(ec, cg[i]), (i, len(T[i]), len(Q[i]))

