"""ctypes bindings used by the tests only.

* ``Ref``    -- the UNMODIFIED reference built by ``oracle/Makefile`` into ``oracle/_ref/libmgref.so``
                (present wherever ``make -C oracle ref`` ran; the built file travels to the GPU box).
* ``Oracle`` -- the plain-C restatement ``oracle/_ref/libmgo.so`` (``make -C oracle restate``).

Nothing in the product imports this module.
"""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmgref.so")
MGO_SO = os.path.join(ROOT, "oracle", "_ref", "libmgo.so")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "minigraph")

m128 = np.dtype([("x", "<u8"), ("y", "<u8")])


class mg128_v(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


class mwf_opt_t(C.Structure):
    _fields_ = [("flag", C.c_int32), ("x", C.c_int32), ("o1", C.c_int32), ("e1", C.c_int32),
                ("o2", C.c_int32), ("e2", C.c_int32), ("step", C.c_int32), ("max_s", C.c_int32),
                ("max_iter", C.c_int64), ("max_occ", C.c_int32), ("kmer", C.c_int32), ("min_len", C.c_int32)]


class mwf_rst_t(C.Structure):
    _fields_ = [("s", C.c_int32), ("n_cigar", C.c_int32), ("n_iter", C.c_int64), ("cigar", C.POINTER(C.c_uint32))]


class mgo_wfa_opt_t(C.Structure):
    _fields_ = [("x", C.c_int32), ("o1", C.c_int32), ("e1", C.c_int32), ("o2", C.c_int32), ("e2", C.c_int32),
                ("max_iter", C.c_int64)]


def have_ref():
    return os.path.exists(REF_SO)


def have_oracle():
    return os.path.exists(MGO_SO)


class Ref:
    """Stage-level entry points of the reference (mgpriv.h:75-126)."""

    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        self.libc = C.CDLL(None)
        L = self.lib
        L.mg_sketch.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(mg128_v)]
        L.mg_sketch.restype = None
        L.radix_sort_128x.argtypes = [C.c_void_p, C.c_void_p]
        L.radix_sort_gfa64.argtypes = [C.c_void_p, C.c_void_p]
        L.mg_lchain_dp.argtypes = [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                                   C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_void_p]
        L.mg_lchain_dp.restype = C.c_void_p
        L.mwf_opt_init.argtypes = [C.POINTER(mwf_opt_t)]
        L.mwf_wfa_auto.argtypes = [C.c_void_p, C.POINTER(mwf_opt_t), C.c_int32, C.c_char_p, C.c_int32, C.c_char_p,
                                   C.POINTER(mwf_rst_t)]
        L.mwf_wfa_auto.restype = None
        L.mwf_wfa_chain.argtypes = L.mwf_wfa_auto.argtypes
        L.mwf_wfa_chain.restype = None
        L.gfa_read.argtypes = [C.c_char_p]
        L.gfa_read.restype = C.c_void_p
        L.gfa_destroy.argtypes = [C.c_void_p]
        L.mg_idx_get.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int)]
        L.mg_idx_get.restype = C.POINTER(C.c_uint64)
        self.libc.free.argtypes = [C.c_void_p]
        self.libc.malloc.argtypes = [C.c_size_t]
        self.libc.malloc.restype = C.c_void_p

    def sketch(self, seq: bytes, w: int, k: int, rid: int = 0) -> np.ndarray:
        v = mg128_v(0, 0, None)
        self.lib.mg_sketch(None, seq, len(seq), w, k, rid, C.byref(v))
        out = np.empty(v.n, dtype=m128)
        if v.n:
            C.memmove(out.ctypes.data, v.a, v.n * 16)
        if v.a:
            self.libc.free(v.a)
        return out

    def sort128x(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a.copy())
        self.lib.radix_sort_128x(a.ctypes.data, a.ctypes.data + a.nbytes)
        return a

    def sort64(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a.astype("<u8").copy())
        self.lib.radix_sort_gfa64(a.ctypes.data, a.ctypes.data + a.nbytes)
        return a

    def lchain_dp(self, a: np.ndarray, max_dist_x=5000, max_dist_y=5000, bw=500, max_skip=25, max_iter=5000,
                  min_cnt=5, min_sc=40, pen_gap=1.0, pen_skip=0.05):
        """mg_lchain_dp frees its input with kfree(km=0) == free(): hand it a malloc'ed copy."""
        n = len(a)
        buf = self.libc.malloc(max(n, 1) * 16)
        C.memmove(buf, a.ctypes.data, n * 16)
        n_u = C.c_int(0)
        u = C.c_void_p()
        r = self.lib.mg_lchain_dp(max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc,
                                  C.c_float(pen_gap), C.c_float(pen_skip), 0, 1, n, buf, C.byref(n_u), C.byref(u), None)
        uu = np.empty(n_u.value, dtype="<u8")
        if n_u.value:
            C.memmove(uu.ctypes.data, u, n_u.value * 8)
        n_a = int((uu & 0xffffffff).sum())
        out = np.empty(n_a, dtype=m128)
        if n_a:
            C.memmove(out.ctypes.data, r, n_a * 16)
        if r:
            self.libc.free(r)
        if u:
            self.libc.free(u)
        return uu, out

    def wfa(self, ts: bytes, qs: bytes):
        opt = mwf_opt_t()
        self.lib.mwf_opt_init(C.byref(opt))
        opt.flag |= 1
        r = mwf_rst_t()
        self.lib.mwf_wfa_auto(None, C.byref(opt), len(ts), ts, len(qs), qs, C.byref(r))
        cig = np.array([r.cigar[i] for i in range(r.n_cigar)], dtype="<u4")
        if r.cigar:
            self.libc.free(r.cigar)
        return r.s, cig

    def wfa_chain(self, ts: bytes, qs: bytes):
        """mwf_wfa_chain() with the options mwf_wfa_auto() (miniwfa.c:829-832) gives it after the exact pass gave up"""
        opt = mwf_opt_t()
        self.lib.mwf_opt_init(C.byref(opt))
        opt.flag |= 1
        opt.step, opt.max_iter = 5000, -1
        r = mwf_rst_t()
        self.lib.mwf_wfa_chain(None, C.byref(opt), len(ts), ts, len(qs), qs, C.byref(r))
        cig = np.ctypeslib.as_array(r.cigar, shape=(r.n_cigar,)).astype("<u4") if r.n_cigar else np.zeros(0, "<u4")
        if r.cigar:
            self.libc.free(r.cigar)
        return r.s, cig


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(MGO_SO)
        L = self.lib
        L.mgo_sketch.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_int64]
        L.mgo_sketch.restype = C.c_int64
        L.mgo_sort128x.argtypes = [C.c_void_p, C.c_int64]
        L.mgo_sort64.argtypes = [C.c_void_p, C.c_int64]
        L.mgo_lchain_dp.argtypes = [C.c_int32] * 7 + [C.c_float, C.c_float, C.c_int64, C.c_void_p, C.c_void_p,
                                                      C.POINTER(C.c_int64)]
        L.mgo_lchain_dp.restype = C.c_int32
        L.mgo_wfa_exact.argtypes = [C.POINTER(mgo_wfa_opt_t), C.c_int32, C.c_char_p, C.c_int32, C.c_char_p,
                                    C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        L.mgo_wfa_exact.restype = C.c_int32
        L.mgo_idx_build.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int32, C.c_int32]
        L.mgo_idx_build.restype = C.c_void_p
        L.mgo_idx_free.argtypes = [C.c_void_p]
        L.mgo_idx_get.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int32)]
        L.mgo_idx_get.restype = C.POINTER(C.c_uint64)
        L.mgo_collect_seed_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]
        L.mgo_collect_seed_hits.restype = C.c_int64

    def sketch(self, seq: bytes, w: int, k: int, rid: int = 0) -> np.ndarray:
        cap = len(seq) + 16
        out = np.empty(cap, dtype=m128)
        n = self.lib.mgo_sketch(seq, len(seq), w, k, rid, out.ctypes.data, cap)
        if n < 0:
            cap = -n
            out = np.empty(cap, dtype=m128)
            n = self.lib.mgo_sketch(seq, len(seq), w, k, rid, out.ctypes.data, cap)
        return out[:n].copy()

    def sort128x(self, a):
        a = np.ascontiguousarray(a.copy())
        self.lib.mgo_sort128x(a.ctypes.data, len(a))
        return a

    def sort64(self, a):
        a = np.ascontiguousarray(a.astype("<u8").copy())
        self.lib.mgo_sort64(a.ctypes.data, len(a))
        return a

    def lchain_dp(self, a, max_dist_x=5000, max_dist_y=5000, bw=500, max_skip=25, max_iter=5000,
                  min_cnt=5, min_sc=40, pen_gap=1.0, pen_skip=0.05):
        a = np.ascontiguousarray(a.copy())
        u = np.zeros(max(len(a), 1), dtype="<u8")
        n_a = C.c_int64(0)
        n_u = self.lib.mgo_lchain_dp(max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc,
                                     C.c_float(pen_gap), C.c_float(pen_skip), len(a), a.ctypes.data, u.ctypes.data,
                                     C.byref(n_a))
        return u[:n_u].copy(), a[:n_a.value].copy()

    def wfa(self, ts: bytes, qs: bytes, max_iter=100000000):
        opt = mgo_wfa_opt_t(4, 4, 2, 15, 1, max_iter)
        cap = len(ts) + len(qs) + 2
        cig = np.zeros(cap, dtype="<u4")
        n = C.c_int32(0)
        it = C.c_int64(0)
        s = self.lib.mgo_wfa_exact(C.byref(opt), len(ts), ts, len(qs), qs, cig.ctypes.data, cap, C.byref(n), C.byref(it))
        return s, cig[:n.value].copy()

    def idx_build(self, segs, w, k):
        n = len(segs)
        arr = (C.c_char_p * n)(*segs)
        lens = (C.c_int32 * n)(*[len(s) for s in segs])
        self._seg_len = np.array([len(s) for s in segs], dtype=np.int32)
        return self.lib.mgo_idx_build(n, arr, lens, w, k)

    def idx_free(self, idx):
        self.lib.mgo_idx_free(idx)

    def seed_hits(self, idx, mz, max_occ):
        mz = np.ascontiguousarray(mz)
        rep, nmp = C.c_int32(0), C.c_int32(0)
        n_a = self.lib.mgo_collect_seed_hits(idx, self._seg_len.ctypes.data, max_occ, len(mz), mz.ctypes.data, None,
                                             C.byref(rep), C.byref(nmp), None)
        a = np.zeros(max(n_a, 1), dtype=m128)
        mp = np.zeros(max(nmp.value, 1), dtype=np.int32)
        n_a = self.lib.mgo_collect_seed_hits(idx, self._seg_len.ctypes.data, max_occ, len(mz), mz.ctypes.data,
                                             a.ctypes.data, C.byref(rep), C.byref(nmp), mp.ctypes.data)
        return a[:n_a].copy(), rep.value, mp[:nmp.value].copy()
