"""CPU: the graph wavefront aligner that bridges two linear chains (minigraph_amd/csrc/gwfa.c) against the reference's own
gfa_ed_init / gfa_ed_step (gfa-ed.c:44-617, called as in gchain1.c:349-381): same edit distance, same vertex path."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import minigraph_amd as mga
import refbind as rb


class edopt_t(C.Structure):
    _fields_ = [("traceback", C.c_int32), ("bw_dyn", C.c_int32), ("max_lag", C.c_int32), ("max_chk", C.c_int32),
                ("s_term", C.c_int32), ("i_term", C.c_int64)]


class edrst_t(C.Structure):
    _fields_ = [("s", C.c_int32), ("end_v", C.c_uint32), ("end_off", C.c_int32), ("wlen", C.c_int32), ("n_end", C.c_int32),
                ("nv", C.c_int32), ("n_iter", C.c_int64), ("v", C.POINTER(C.c_int32))]


COMP = bytes.maketrans(b"ACGT", b"TGCA")


def load_gfa(path):
    segs, names, arcs = [], {}, {}
    for line in open(path, "rb"):
        f = line.rstrip(b"\n").split(b"\t")
        if f[0] == b"S":
            names[f[1]] = len(segs)
            segs.append(f[2].upper())
    for line in open(path, "rb"):
        f = line.rstrip(b"\n").split(b"\t")
        if f[0] == b"L":
            v = names[f[1]] << 1 | (f[2] == b"-")
            w = names[f[3]] << 1 | (f[4] == b"-")
            arcs.setdefault(v, []).append(w)
            arcs.setdefault(w ^ 1, []).append(v ^ 1)
    return segs, arcs


def vseq(segs, v):
    s = segs[v >> 1]
    return s.translate(COMP)[::-1] if v & 1 else s


def mutate(rng, s, rate):
    out = bytearray()
    for c in s:
        r = rng.random()
        if r < rate / 3:
            out.append(int(rng.choice([x for x in b"ACGT" if x != c])))
        elif r < 2 * rate / 3:
            continue
        elif r < rate:
            out.append(c)
            out.append(int(rng.choice(list(b"ACGT"))))
        else:
            out.append(c)
    return bytes(out)


def sample_walk(rng, segs, arcs, length):
    """a walk of about `length` bases: (v0, off0) .. (v1, off1), both ends inclusive"""
    v = int(rng.integers(0, 2 * len(segs)))
    s = vseq(segs, v)
    off0 = int(rng.integers(0, max(1, len(s) - 20)))
    if rng.random() < 0.75:  # most walks start close to a vertex end, so that they run through bubbles
        off0 = max(0, len(s) - int(rng.integers(10, max(11, length))))
    v0, seq, path = v, s[off0:], [v]
    while len(seq) < length and arcs.get(v):
        v = int(rng.choice(arcs[v]))
        path.append(v)
        seq += vseq(segs, v)
    over = max(0, len(seq) - length)
    last_len = len(vseq(segs, path[-1]))
    if over >= last_len:
        over = last_len - 1
    if len(path) == 1:
        over = min(over, len(seq) - 1)
    seq = seq[:len(seq) - over]
    off1 = last_len - over - 1
    if len(path) == 1:
        off1 = off0 + len(seq) - 1
    return v0, off0, path[-1], off1, seq, path


@pytest.mark.skipif(not rb.have_oracle(), reason="oracle/_ref not built")
@pytest.mark.parametrize("rate,length,s_term", [(0.05, 400, 10000), (0.15, 1200, 10000), (0.25, 300, 10000), (0.12, 3000, 10000), (0.2, 600, 25)])
def test_gwfa_bridge_matches_reference(rate, length, s_term):
    L, R = mga.load(), rb.Ref().lib
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t"), "-G", "400000", "-H", "4", "-n", "1", "-s", "13"], stderr=subprocess.DEVNULL)
    gfa = os.path.join(d, "t.gfa")
    segs, arcs = load_gfa(gfa)
    # ours
    io, mo, go = mga.idxopt_t(), mga.mapopt_t(), mga.ggopt_t()
    L.mg_opt_set(None, C.byref(io), C.byref(mo), C.byref(go))
    L.gfa_read.restype = C.c_void_p
    L.mga_idx_hostpart.restype = C.c_void_p
    L.mga_idx_hostpart.argtypes = [C.c_void_p, C.c_void_p]
    g = L.gfa_read(gfa.encode())
    gi = L.mga_idx_hostpart(g, C.byref(io))
    es = C.cast(gi, C.POINTER(C.c_void_p))[1]                   # mg_idx_t.es (minigraph.h:93-98)
    L.mga_gwfa_bridge.restype = C.c_int32
    L.mga_gwfa_bridge.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_char_p, C.c_uint32, C.c_int32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32,
                                  C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int32)]
    # reference
    R.gfa_read.restype = C.c_void_p
    R.gfa_edseq_init.restype = C.c_void_p
    R.gfa_edseq_init.argtypes = [C.c_void_p]
    R.gfa_ed_init.restype = C.c_void_p
    R.gfa_ed_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_char_p, C.c_uint32, C.c_int32]
    R.gfa_ed_step.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p]
    R.gfa_ed_destroy.argtypes = [C.c_void_p]
    gr = R.gfa_read(gfa.encode())
    esr = R.gfa_edseq_init(gr)
    opt = edopt_t()
    R.gfa_edopt_init(C.byref(opt))
    opt.traceback, opt.max_chk, opt.bw_dyn, opt.max_lag, opt.i_term = 1, 1000, 1000, 5000, 500000000
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(int(rate * 1000) + length)
    n_found = n_multi = 0
    for it in range(40):
        v0, off0, v1, off1, seq, path = sample_walk(rng, segs, arcs, length)
        q = mutate(rng, seq, rate)
        if len(q) < 2:
            continue
        p_ours, nv = C.POINTER(C.c_int32)(), C.c_int32(0)
        s_ours = L.mga_gwfa_bridge(g, es, len(q), q, v0, off0, v1, off1, 5000, s_term, C.byref(p_ours), C.byref(nv))
        r = edrst_t()
        z = R.gfa_ed_init(None, C.byref(opt), gr, esr, len(q), q, v0, off0)
        R.gfa_ed_step(z, v1, off1, s_term, C.byref(r))
        R.gfa_ed_destroy(z)
        assert s_ours == r.s, (it, s_ours, r.s, len(q), len(path))
        if r.s >= 0:
            n_found += 1
            n_multi += r.nv > 1
            assert nv.value == r.nv, it
            assert [p_ours[i] for i in range(nv.value)] == [r.v[i] for i in range(r.nv)], it
            libc.free(r.v)
        if p_ours:
            libc.free(p_ours)
    if s_term > 1000:
        assert n_found >= 30 and n_multi >= 12, (n_found, n_multi)
