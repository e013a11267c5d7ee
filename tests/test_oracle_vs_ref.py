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
