"""CPU: the host RMQ chainer (minigraph_amd/csrc/rmq.c: primary chainer under -x asm, long-join rescue fallback under -x lr)
against the reference's own mg_lchain_rmq() (lchain.c:252-372) on synthetic anchor sets: same chains, same compacted anchors."""
import ctypes as C

import numpy as np
import pytest

import minigraph_amd as mga
import refbind as rb

M128 = np.dtype([("x", "<u8"), ("y", "<u8")])


def anchors(rng, n, kind):
    gaps = rng.integers(1, 12, n).astype(np.int64)
    x = np.cumsum(gaps) + 17
    if kind == "colinear":       # one long diagonal with small indels
        y = x + 100 + np.cumsum((rng.random(n) < 0.002) * rng.integers(-30, 30, n))
    elif kind == "noisy":        # plus off-diagonal noise hits and repeated target positions (equal x)
        y = x + 100 + np.cumsum((rng.random(n) < 0.01) * rng.integers(-200, 200, n))
        noise = rng.random(n) < 0.1
        y = np.where(noise, rng.integers(20, x[-1] + 1000, n), y)
        dup = rng.random(n) < 0.05
        x = np.where(dup, np.roll(x, 1), x)
        x[0] = 17
        x = np.sort(x)
    elif kind == "dense":        # anchors every few bases, half of them off the diagonal by up to 400: every inner walk visits dozens of keys, many of the outer window only
        x = np.sort(rng.integers(17, 4 * n, n))
        y = x + 100 + np.where(rng.random(n) < 0.5, rng.integers(-400, 400, n), 0)
    else:                        # "jumps": long gaps on either axis, several target segments
        y = x + 100 + np.cumsum((rng.random(n) < 0.003) * rng.integers(-8000, 8000, n))
        x = x + np.cumsum((rng.random(n) < 0.002) * rng.integers(0, 30000, n))
        seg = np.cumsum(rng.random(n) < 0.0005).astype(np.int64)
        x = x + (seg << 33)
    y = np.clip(y, 17, None)
    a = np.zeros(n, dtype=M128)
    a["x"] = x.astype(np.uint64)
    a["y"] = (np.uint64(17) << np.uint64(32)) | y.astype(np.uint64)
    return a[np.argsort(a["x"], kind="stable")]


def run_ours(L, a, par):
    L.mga_lchain_rmq.restype = C.c_void_p
    L.mga_lchain_rmq.argtypes = [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int64, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
    nu, u = C.c_int(0), C.c_void_p()
    r = L.mga_lchain_rmq(*par, 0.8, 0.05, len(a), a.ctypes.data, C.byref(nu), C.byref(u))
    return take(r, u, nu.value)


def take(r, u, nu):
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    uu = np.ctypeslib.as_array(C.cast(u, C.POINTER(C.c_uint64)), shape=(nu,)).copy() if nu else np.zeros(0, np.uint64)
    n_b = int((uu & np.uint64(0xffffffff)).sum())
    b = np.frombuffer(C.string_at(r, n_b * 16), dtype=M128).copy() if n_b else np.zeros(0, M128)
    libc.free(r)
    libc.free(u)
    return uu, b


def run_ref(R, a, par):
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    buf = libc.malloc(len(a) * 16)  # the reference frees its input
    C.memmove(buf, a.ctypes.data, len(a) * 16)
    R.mg_lchain_rmq.restype = C.c_void_p
    R.mg_lchain_rmq.argtypes = [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int64, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_void_p]
    nu, u = C.c_int(0), C.c_void_p()
    r = R.mg_lchain_rmq(*par, 0.8, 0.05, len(a), buf, C.byref(nu), C.byref(u), None)
    return take(r, u, nu.value)


@pytest.mark.parametrize("kind", ["colinear", "noisy", "jumps", "dense"])
@pytest.mark.parametrize("par", [(10000, 1000, 2000, 25, 100000, 5, 40),   # -x asm shape
                                 (5000, 1000, 20000, 25, 100000, 5, 40),   # the -x lr rescue (bw_long)
                                 (3000, 0, 500, 5, 300, 3, 20),            # no inner tree, a tree cap that bites, few skips
                                 (10000, 1000, 2000, 25, 60, 5, 40),       # a cap that bites BOTH windows (the inner window's size clause, lchain.c:304)
                                 (4000, 3000, 2000, 3, 200, 5, 40),        # inner window nearly the outer one, cap between them, few skips
                                 (10000, 150, 2000, 25, 100000, 5, 40)])   # a narrow inner window: its walk passes over many keys of the outer window only (one tree, rmq.c)
def test_rmq_chainer_matches_reference(kind, par):
    L, R = mga.load(), rb.Ref().lib
    for seed in range(3):
        rng = np.random.default_rng(100 * seed + len(kind))
        a = anchors(rng, int(rng.integers(2000, 40000)), kind)
        u1, b1 = run_ours(L, a, par)
        u2, b2 = run_ref(R, a, par)
        assert np.array_equal(u1, u2), (kind, par, seed)
        assert np.array_equal(b1, b2), (kind, par, seed)
        assert len(u1) > 0


def test_host_radix_sort_single_and_pool_equal_the_reference_permutation():
    """ksortx.c: the product's host twin of klib's in-place MSD radix sort (ksort.h:112-162, radix_sort_128x): same PERMUTATION as the reference's -- equal keys included --
    on one thread and fanned out over the pool (round 5: the anchor sort and the backtrack's (score, index) sort of a chromosome-scale contig)"""
    import ctypes as C
    import numpy as np
    import minigraph_amd as mga
    ref = rb.Ref()
    L = mga.load()
    L.mga_ksort_128x.argtypes = [C.c_int64, C.c_void_p]
    thr = C.c_int.in_dll(L, "mga_ksort_threads")
    rng = np.random.default_rng(5)
    for n, key_bits in ((300_000, 12), (300_000, 30), (700_000, 52), (1_200_000, 20), (1_200_000, 40), (400_000, 64)):
        a = np.zeros(n, dtype=rb.m128)
        a["x"] = rng.integers(0, 2 ** min(key_bits, 63), size=n, dtype=np.uint64)
        if key_bits == 52:   # anchor-like keys: a few (segment, strand) groups, positions inside them -- heavily skewed first passes
            a["x"] = (rng.integers(0, 9, size=n, dtype=np.uint64) << np.uint64(33)) | rng.integers(0, 2 ** 22, size=n, dtype=np.uint64)
        if key_bits == 64:
            a["x"] |= rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63)
        a["y"] = np.arange(n, dtype=np.uint64)   # the payload exposes the order of equal keys
        want = ref.sort128x(a)
        for t in (1, 3, 16):
            thr.value = t
            got = a.copy()
            L.mga_ksort_128x(n, got.ctypes.data)
            assert np.array_equal(got, want), (n, key_bits, t)
    thr.value = 1
