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


@pytest.mark.parametrize("w,k", [(11, 17), (10, 19), (10, 21), (5, 4), (3, 6), (1, 5), (16, 28), (200, 15), (255, 28)])
def test_sketch_parity(ora, w, k):
    rng = np.random.default_rng(1000 + w * 31 + k)
    seqs = []
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


def test_sketch_many_reads(ora):
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
