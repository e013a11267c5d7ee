"""Test harness (CPU): drive the PRODUCT's host halves (mga_batch_chain / mga_batch_finish) with stage
inputs computed by the ORACLE restatement, so that the host logic is checked without a GPU.
The oracle stands in for the HIP kernels here ONLY inside tests."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

import minigraph_amd as mga
import refbind as rb


class wfa_prob_t(C.Structure):
    _fields_ = [("t_off", C.c_int64), ("q_off", C.c_int64), ("tl", C.c_int32), ("ql", C.c_int32)]


class wfa_res_t(C.Structure):
    _fields_ = [("score", C.c_int32), ("n_cigar", C.c_int32), ("cig_off", C.c_int64), ("status", C.c_int32),
                ("pad", C.c_int32), ("n_iter", C.c_int64)]


class kstring_t(C.Structure):
    _fields_ = [("l", C.c_uint), ("m", C.c_uint), ("s", C.c_void_p)]


def read_fa(path):
    names, seqs, cur = [], [], []
    for line in open(path, "rb"):
        if line.startswith(b">"):
            if names:
                seqs.append(b"".join(cur).upper().replace(b"U", b"T"))
            names.append(line[1:].split()[0])
            cur = []
        else:
            cur.append(line.strip())
    if names:
        seqs.append(b"".join(cur).upper().replace(b"U", b"T"))
    return names, seqs


def graph_segments(path):
    if open(path, "rb").read(1) == b">":
        return read_fa(path)[1]
    return [l.split(b"\t")[2].upper() for l in open(path, "rb") if l.startswith(b"S\t")]


def run_reference(graph, reads, cigar=True, threads=4, preset="lr"):
    args = [rb.REF_BIN] + (["-c"] if cigar else []) + ["-x", preset, "-t", str(threads), graph, reads]
    p = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    m = re.search(r"occ_max1=(\d+); lc_max_occ=(\d+)", p.stderr.decode())
    return p.stdout, int(m.group(1)), int(m.group(2))


def map_with_oracle_stages(graph, reads, occ_max1, lc_max_occ, cigar=True, n_threads=4, preset="lr", per_read=False, return_chains=False, host_dp=False):
    """whole -cx lr (or -cx asm) job: oracle for the kernel stages, product C code for everything on the host.  Under asm the RMQ chainer
    is the primary chainer and runs in the product's host phases (mapper.c: rq_chain_all) on the oracle's sorted anchors"""
    L = mga.load()
    ora = rb.Oracle()
    pp = C.POINTER(C.c_void_p)
    L.mga_idx_hostpart.argtypes = [C.c_void_p, C.POINTER(mga.idxopt_t)]
    L.mga_idx_hostpart.restype = C.c_void_p
    L.mga_batch_init.argtypes = [C.c_void_p, C.POINTER(mga.mapopt_t), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.mga_batch_init.restype = C.c_void_p
    L.mga_batch_chain.argtypes = [C.c_void_p] + [C.c_void_p] * 9 + [C.c_int, C.c_void_p]
    L.mga_batch_n_wfa.argtypes = [C.c_void_p]
    L.mga_batch_n_wfa.restype = C.c_int64
    L.mga_batch_wfa_target_bytes.argtypes = [C.c_void_p]
    L.mga_batch_wfa_target_bytes.restype = C.c_int64
    L.mga_batch_wfa_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mga_batch_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mga_batch_take_results.argtypes = [C.c_void_p]
    L.mga_batch_take_results.restype = C.POINTER(C.c_void_p)
    L.mga_batch_destroy.argtypes = [C.c_void_p]
    L.mga_batch_lchain_par.argtypes = [C.c_void_p, C.POINTER(mga.mapopt_t), C.c_int, C.POINTER(mga.lchain_par_t)]
    L.mg_write_gaf.argtypes = [C.POINTER(kstring_t), C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_char_p, C.c_uint64, C.c_void_p]
    L.mg_gchain_free.argtypes = [C.c_void_p]

    io, mo, go = mga.idxopt_t(), mga.mapopt_t(), mga.ggopt_t()
    L.mg_opt_set(None, C.byref(io), C.byref(mo), C.byref(go))
    L.mg_opt_set(preset.encode(), C.byref(io), C.byref(mo), C.byref(go))
    is_rmq = 1 if (mo.flag & 0x8000) else 0
    if cigar:
        mo.flag |= mga.MG_M_CIGAR
    mo.occ_max1, mo.lc_max_occ = occ_max1, lc_max_occ  # what mg_opt_update derives from the index
    g = L.gfa_read(graph.encode())
    gi = L.mga_idx_hostpart(g, C.byref(io))
    names, seqs = read_fa(reads)
    n = len(seqs)
    # ---- kernel stages through the oracle ----
    oidx = ora.idx_build(graph_segments(graph), io.w, io.k)
    par = mga.lchain_par_t()
    L.mga_batch_lchain_par(gi, C.byref(mo), 0, C.byref(par))
    n_mz, rep, minis, nus, nbs, us, aas = [], [], [], [], [], [], []
    for s in seqs:
        mz = ora.sketch(s, io.w, io.k)
        a, rl, mp = ora.seed_hits(oidx, mz, occ_max1)
        if is_rmq or host_dp:
            u, b = np.zeros(0, dtype=np.uint64), a  # raw x-sorted anchors: the host chains (host_dp: the placement of ultra-long -x lr reads -- first pass mga_lchain_dp_fwd, rescue by the RMQ tree)
        else:
            u, b = ora.lchain_dp(a, max_dist_x=par.max_dist_x, max_dist_y=par.max_dist_y, bw=par.bw, max_skip=par.max_skip,
                                 max_iter=par.max_iter, min_cnt=par.min_cnt, min_sc=par.min_sc, pen_gap=par.chn_pen_gap, pen_skip=par.chn_pen_skip)
        n_mz.append(len(mz)); rep.append(rl); minis.append(mp)
        nus.append(len(u)); nbs.append(len(b)); us.append(u); aas.append((len(a), b))
    ora.idx_free(oidx)
    a_off = np.zeros(n + 1, dtype=np.int64)
    a_off[1:] = np.cumsum([x[0] for x in aas])
    U = np.zeros(int(a_off[-1]) + 1, dtype=np.uint64)
    A = np.zeros(int(a_off[-1]) + 1, dtype=mga.m128)
    for i in range(n):
        U[a_off[i]:a_off[i] + nus[i]] = us[i]
        A[a_off[i]:a_off[i] + nbs[i]] = aas[i][1]
    mini_off = np.zeros(n + 1, dtype=np.int64)
    mini_off[1:] = np.cumsum([len(m) for m in minis])
    MINI = np.ascontiguousarray(np.concatenate(minis + [np.zeros(1, dtype=np.int32)]), dtype=np.int32)
    qlens = (C.c_int * n)(*[len(s) for s in seqs])
    seqp = (C.c_char_p * n)(*seqs)
    namep = (C.c_char_p * n)(*names)
    q_off = np.zeros(n + 1, dtype=np.int64)
    q_off[1:] = np.cumsum([len(s) for s in seqs])
    qcat = b"".join(seqs)
    i32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
    N_MZ, REP, NU, NB = i32(n_mz), i32(rep), i32(nus), i32(nbs)
    b = L.mga_batch_init(gi, C.byref(mo), n, qlens, seqp, namep, q_off.ctypes.data, n_threads)
    assert L.mga_batch_chain(b, N_MZ.ctypes.data, REP.ctypes.data, MINI.ctypes.data, mini_off.ctypes.data, NU.ctypes.data, NB.ctypes.data,
                             U.ctypes.data, A.ctypes.data, a_off.ctypes.data, 5 if (host_dp and not is_rmq) else is_rmq, None) == 0
    n_prob, n_tb = L.mga_batch_n_wfa(b), L.mga_batch_wfa_target_bytes(b)
    probs = (wfa_prob_t * max(n_prob, 1))()
    tbuf = C.create_string_buffer(int(n_tb) + 64)
    res = (wfa_res_t * max(n_prob, 1))()
    pool = []
    if n_prob:
        L.mga_batch_wfa_export(b, probs, tbuf)
        traw = tbuf.raw
        for j in range(n_prob):
            p = probs[j]
            s, cg = ora.wfa(traw[p.t_off:p.t_off + p.tl], qcat[p.q_off:p.q_off + p.ql])
            res[j].score, res[j].n_cigar, res[j].cig_off, res[j].status = s, len(cg), len(pool), 0
            pool.extend(int(x) for x in cg)
    POOL = np.ascontiguousarray(np.array(pool + [0], dtype=np.uint32))
    assert L.mga_batch_finish(b, res, POOL.ctypes.data) == 0
    gcs = L.mga_batch_take_results(b)
    if return_chains:   # the mg_gchains_t* themselves (the caller frees them with mg_gchain_free): what ggen_map hands to --call / graph generation
        L.mga_batch_destroy(b)
        return dict(gcs=gcs, n=n, names=names, seqs=seqs, flag=mo.flag)
    ks = kstring_t(0, 0, None)
    out = []
    for i in range(n):
        ql = C.c_int32(len(seqs[i]))
        L.mg_write_gaf(C.byref(ks), g, gcs[i], 1, C.byref(ql), names[i], mo.flag, None)
        if ks.l or per_read:
            out.append(C.string_at(ks.s, ks.l) if ks.l else b"")
        L.mg_gchain_free(gcs[i])
    L.mga_batch_destroy(b)
    return (out if per_read else b"".join(out)), n_prob
