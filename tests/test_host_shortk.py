"""CPU: the k-shortest-walk search between chain ends (minigraph_amd/csrc/shortk.c) against the reference's mg_shortest_k()
(shortk.c:41-242) on a bubble graph: same per-destination results (distance, walk hash, n_path, is_0) and the same walk vertices."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import minigraph_amd as mga
import refbind as rb
from test_host_gwfa import load_gfa


class path_dst_t(C.Structure):  # mgpriv.h:40-52
    _fields_ = [("v", C.c_uint32), ("target_dist", C.c_int32), ("target_hash", C.c_uint32), ("meta_flags", C.c_uint32), ("qlen", C.c_int32),
                ("n_path_is0", C.c_uint32), ("path_end", C.c_int32), ("dist", C.c_int32), ("hash", C.c_uint32)]


def reachable(rng, segs, arcs, src, max_steps):
    out, v, dist = [], src, 0
    for _ in range(max_steps):
        if not arcs.get(v):
            break
        w = int(rng.choice(arcs[v]))
        out.append((w, dist))            # distance between the end of src and the start of w
        dist += len(segs[w >> 1])
        v = w
    return out


@pytest.mark.skipif(not rb.have_oracle(), reason="oracle/_ref not built")
@pytest.mark.parametrize("max_dist,max_k", [(20000, 15), (3000, 15), (100000, 3)])
def test_shortest_k_matches_reference(max_dist, max_k):
    L, R = mga.load(), rb.Ref().lib
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "600000", "-H", "5", "-n", "1", "-s", "17"], stderr=subprocess.DEVNULL)
    gfa = os.path.join(d, "t.gfa")
    segs, arcs = load_gfa(gfa)
    L.gfa_read.restype = C.c_void_p
    R.gfa_read.restype = C.c_void_p
    g, gr = L.gfa_read(gfa.encode()), R.gfa_read(gfa.encode())
    L.mga_shortest_k.restype = C.c_void_p
    L.mga_shortest_k.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    R.mg_shortest_k.restype = C.c_void_p
    R.mg_shortest_k.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(max_dist + max_k)
    n_with_path = 0
    for it in range(60):
        src = int(rng.integers(0, 2 * len(segs)))
        cand = reachable(rng, segs, arcs, src, int(rng.integers(1, 9)))
        if not cand:
            continue
        picks = [cand[i] for i in sorted(set(int(x) for x in rng.integers(0, len(cand), size=int(rng.integers(1, 4)))))]
        if rng.random() < 0.2:  # a destination that cannot be reached going forward
            picks.append((src ^ 1, 1000))
        for phase in (0, 1):
            n = len(picks)
            A, B = (path_dst_t * n)(), (path_dst_t * n)()
            for i, (w, dist) in enumerate(picks):
                for X in (A, B):
                    X[i].v, X[i].qlen = w, 0
                    if phase == 0:
                        X[i].target_dist, X[i].target_hash, X[i].meta_flags = dist + int(rng.integers(-50, 50)) if i % 2 else -1, 0, 0
                        X[i].target_dist = A[i].target_dist
                    else:  # what bridge_shortk does (gchain1.c:319-347): ask for the walk found before, by distance and hash
                        X[i].target_dist, X[i].target_hash, X[i].meta_flags = prev[i][0], prev[i][1], 1 << 30
            na, nb = C.c_int32(0), C.c_int32(0)
            pa = L.mga_shortest_k(g, src, n, A, max_dist, max_k, C.byref(na))
            pb = R.mg_shortest_k(None, gr, src, n, B, max_dist, max_k, C.byref(nb))
            assert bytes(A) == bytes(B), (it, phase)
            assert na.value == nb.value, (it, phase)
            if na.value:
                assert C.string_at(pa, na.value * 12) == C.string_at(pb, nb.value * 12), (it, phase)
                n_with_path += 1
            prev = [(A[i].dist, A[i].hash) for i in range(n)]
            libc.free(pa)
            libc.free(pb)
    assert n_with_path >= 20
