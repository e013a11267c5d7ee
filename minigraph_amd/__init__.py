"""minigraph_amd -- thin ctypes loader for libminigraph_amd.so (the C-ABI product library).

Python is plumbing here (tests, bench harness); the product is the shared library built from
``minigraph_amd/csrc`` (host C + hand-written HIP kernels for gfx950).  There is NO CPU fallback:
if the library is missing, ``load()`` raises, and every entry point fails when no GPU is visible.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_PATH = os.path.join(PKG, "lib", "libminigraph_amd.so")
MGSIM = os.path.join(PKG, "lib", "mgsim")

m128 = np.dtype([("x", "<u8"), ("y", "<u8")])

_lib = None


def build(verbose=False):
    """Compile the library (hipcc --offload-arch=gfx950 + gcc) and the workload generator in-tree."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-j8"], stdout=out)
    os.makedirs(os.path.join(PKG, "lib"), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-o", MGSIM, os.path.join(PKG, "tools", "mgsim.c")], stdout=out)


class lchain_par_t(C.Structure):
    _fields_ = [("max_dist_x", C.c_int32), ("max_dist_y", C.c_int32), ("bw", C.c_int32), ("max_skip", C.c_int32),
                ("max_iter", C.c_int32), ("min_cnt", C.c_int32), ("min_sc", C.c_int32),
                ("chn_pen_gap", C.c_float), ("chn_pen_skip", C.c_float)]


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the HIP path)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    pp = C.POINTER(C.c_void_p)
    L.mga_free.argtypes = [C.c_void_p]
    L.mga_device_count.restype = C.c_int
    L.mga_last_error.restype = C.c_char_p
    L.mga_sketch_batch.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, pp, pp]
    L.mga_wfa_batch.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_char_p, C.c_void_p, pp, pp, pp]
    _lib = L
    return L


def _take(ptr, n, dtype):
    """copy n items out of a malloc'ed C buffer and release it"""
    L = load()
    out = np.empty(n, dtype=dtype)
    if n and ptr.value:
        C.memmove(out.ctypes.data, ptr.value, out.nbytes)
    if ptr.value:
        L.mga_free(ptr)
    return out


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, load().mga_last_error().decode()))


def concat(seqs):
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs])
    return b"".join(seqs), off


def sketch_batch(seqs, w, k, rid=None):
    """mg_sketch for a list of byte strings -> list of m128 arrays (HIP kernel k_sketch)."""
    L = load()
    buf, off = concat(seqs)
    mz, mzo = C.c_void_p(), C.c_void_p()
    r = None if rid is None else np.ascontiguousarray(rid, dtype=np.uint32)
    _check(L.mga_sketch_batch(len(seqs), buf, off.ctypes.data, None if r is None else r.ctypes.data, w, k,
                              C.byref(mz), C.byref(mzo)), "mga_sketch_batch")
    o = _take(mzo, len(seqs) + 1, np.int64)
    a = _take(mz, int(o[-1]), m128)
    return [a[o[i]:o[i + 1]] for i in range(len(seqs))]


def wfa_batch(targets, queries):
    """exact miniwfa alignment of n (target, query) pairs -> (scores, list of uint32 cigar arrays)."""
    L = load()
    tb, to = concat(targets)
    qb, qo = concat(queries)
    sc, cg, co = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _check(L.mga_wfa_batch(len(targets), tb, to.ctypes.data, qb, qo.ctypes.data, C.byref(sc), C.byref(cg), C.byref(co)),
           "mga_wfa_batch")
    n = len(targets)
    o = _take(co, n + 1, np.int64)
    s = _take(sc, n, np.int32)
    c = _take(cg, int(o[-1]), np.uint32)
    return s, [c[o[i]:o[i + 1]] for i in range(n)]
