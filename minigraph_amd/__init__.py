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
LIB_PATH = os.environ.get("MGA_LIB") or os.path.join(PKG, "lib", "libminigraph_amd.so")  # MGA_LIB: a test build of the same library (tests/test_wave_model.py)
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


class idxopt_t(C.Structure):
    _fields_ = [("w", C.c_int), ("k", C.c_int), ("bucket_bits", C.c_int)]


class mapopt_t(C.Structure):  # minigraph.h:51-77
    _fields_ = [("flag", C.c_uint64), ("mini_batch_size", C.c_int64), ("seed", C.c_int), ("max_qlen", C.c_int),
                ("pe_ori", C.c_int), ("occ_max1", C.c_int), ("occ_max1_cap", C.c_int), ("occ_max1_frac", C.c_float),
                ("bw", C.c_int), ("bw_long", C.c_int), ("rmq_size_cap", C.c_int), ("rmq_rescue_size", C.c_int),
                ("rmq_rescue_ratio", C.c_float), ("max_gap_pre", C.c_int), ("max_gap", C.c_int),
                ("max_gap_ref", C.c_int), ("max_frag_len", C.c_int), ("div", C.c_float), ("chn_pen_gap", C.c_float),
                ("chn_pen_skip", C.c_float), ("max_lc_skip", C.c_int), ("max_lc_iter", C.c_int),
                ("max_gc_skip", C.c_int), ("min_lc_cnt", C.c_int), ("min_lc_score", C.c_int), ("min_gc_cnt", C.c_int),
                ("min_gc_score", C.c_int), ("gdp_max_ed", C.c_int), ("lc_max_trim", C.c_int), ("lc_max_occ", C.c_int),
                ("mask_level", C.c_float), ("sub_diff", C.c_int), ("best_n", C.c_int), ("pri_ratio", C.c_float),
                ("ref_bonus", C.c_int), ("cap_kalloc", C.c_int64), ("min_cov_mapq", C.c_int), ("min_cov_blen", C.c_int)]


class ggopt_t(C.Structure):
    _fields_ = [("flag", C.c_uint64), ("algo", C.c_int), ("min_mapq", C.c_int), ("min_map_len", C.c_int),
                ("min_depth_len", C.c_int), ("min_var_len", C.c_int), ("match_pen", C.c_int),
                ("ggs_shrink_pen", C.c_int), ("ggs_min_end_cnt", C.c_int), ("ggs_min_end_frac", C.c_float),
                ("ggs_max_iden", C.c_float), ("ggs_min_inv_iden", C.c_float)]


MG_M_CIGAR = 0x4000000

KERNELS = ["k_sketch", "k_seed_count", "k_seed_fill", "k_lchain", "k_wfa_r[64]", "k_wfa_r[128]", "k_wfa_r[192]", "k_wfa_r[256]", "k_wfa_r[512]",
           "k_wfa_r[1024]", "k_wfa_r[2048]", "k_wfa[hbm4096]", "k_wfa[hbm32768]", "k_scan", "k_text", "k_gchain", "k_plan",
           "k_wfa_w[16x4]", "k_wfa_w[32x2]", "k_wfa_w[64]", "k_wfa_w[128]", "k_wfa_w[192]", "k_wfa_w[256]", "k_wfa_tb", "k_gchain_p2", "k_gchain_p3", "k_gaf"]


class stats_t(C.Structure):  # mga_stats_t
    _fields_ = [(n, C.c_int64) for n in ("n_reads", "n_bases", "n_mz", "n_probe", "n_hit", "n_anchor_chained", "n_wfa",
                                         "wfa_t_bases", "wfa_q_bases", "wfa_cells", "gaf_bytes")] + \
               [(n, C.c_double) for n in ("t_sketch", "t_seed", "t_lchain", "t_host_chain", "t_wfa", "t_host_post", "t_gaf")] + \
               [(n, C.c_int64) for n in ("n_rescue_dev", "n_rescue_host", "n_gwfa", "n_shortk", "n_gc_retry", "gc_arena_peak", "n_wfa_dev_plan", "n_gc_host")]


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
    L.gfa_read.argtypes = [C.c_char_p]
    L.gfa_read.restype = C.c_void_p
    L.gfa_destroy.argtypes = [C.c_void_p]
    L.mg_opt_set.argtypes = [C.c_char_p, C.POINTER(idxopt_t), C.POINTER(mapopt_t), C.POINTER(ggopt_t)]
    L.mg_index.argtypes = [C.c_void_p, C.POINTER(idxopt_t), C.c_int, C.POINTER(mapopt_t)]
    L.mg_index.restype = C.c_void_p
    L.mg_idx_destroy.argtypes = [C.c_void_p]
    L.mga_graph_image_save.argtypes = [C.c_void_p, C.c_char_p]
    L.mga_index_load_image.argtypes = [C.c_char_p, C.POINTER(idxopt_t), C.c_int, C.POINTER(mapopt_t)]
    L.mga_index_load_image.restype = C.c_void_p
    L.mga_seed_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, pp, pp, pp, pp, pp]
    L.mga_lchain_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(lchain_par_t), pp, pp, pp, pp]
    L.mga_map_files_to_path.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(idxopt_t),
                                        C.POINTER(mapopt_t), C.c_int, C.c_char_p]
    L.mga_map_files_shard.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(mapopt_t), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      pp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), pp, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.mga_reads_shard_dump.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_char_p, pp, C.POINTER(C.c_int)]
    L.mga_reads_load.argtypes = [C.c_char_p, C.c_int64]
    L.mga_reads_load.restype = C.c_void_p
    L.mga_reads_free.argtypes = [C.c_void_p]
    L.mga_reads_count.argtypes = [C.c_void_p]
    L.mga_reads_bases.argtypes = [C.c_void_p]
    L.mga_reads_bases.restype = C.c_int64
    L.mga_map_reads.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(mapopt_t), C.c_int, pp, C.POINTER(C.c_int64)]
    L.mga_get_stats.argtypes = [C.c_void_p, C.POINTER(stats_t), C.c_int]
    L.mga_prof_enable.argtypes = [C.c_int]
    L.mga_idx_stream_close.argtypes = [C.c_void_p]
    L.mga_prof_get.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
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


class Graph:
    """gfa_read() + mg_index(): the graph and its minimizer index (host + HBM replica)."""

    def __init__(self, path, preset="lr", cigar=True, n_threads=4, image=False):
        """image=True: `path` is a graph image written by save_image() (mga_graph_image_save): mapped, not parsed; the index owns the graph"""
        L = load()
        self.io, self.mo, self.go = idxopt_t(), mapopt_t(), ggopt_t()
        L.mg_opt_set(None, C.byref(self.io), C.byref(self.mo), C.byref(self.go))
        if L.mg_opt_set(preset.encode(), C.byref(self.io), C.byref(self.mo), C.byref(self.go)) != 0:
            raise ValueError("unknown preset %r" % preset)
        if cigar:
            self.mo.flag |= MG_M_CIGAR
        if image:
            self.g = None
            self.gi = L.mga_index_load_image(path.encode(), C.byref(self.io), n_threads, C.byref(self.mo))
            if not self.gi:
                raise RuntimeError("mga_index_load_image(%s) failed: %s" % (path, L.mga_last_error().decode()))
            return
        self.g = L.gfa_read(path.encode())
        if not self.g:
            raise RuntimeError("gfa_read(%s) failed" % path)
        self.gi = L.mg_index(self.g, C.byref(self.io), n_threads, C.byref(self.mo))
        if not self.gi:
            raise RuntimeError("mg_index failed: %s" % L.mga_last_error().decode())

    def save_image(self, path):
        """the graph as one binary image (mga_graph_image_save): Graph(path, image=True) maps it instead of parsing GFA text"""
        if not self.g:
            raise RuntimeError("this graph was loaded from an image")
        _check(load().mga_graph_image_save(self.g, path.encode()), "mga_graph_image_save")

    def close(self):
        L = load()
        if self.gi:
            L.mg_idx_destroy(self.gi)
        if self.g:
            L.gfa_destroy(self.g)
        self.gi = self.g = None

    def seed_batch(self, mz_list, max_occ=None):
        """collect_seed_hits for each read's minimizers -> list of (anchors, rep_len, mini_pos)"""
        L = load()
        n = len(mz_list)
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(m) for m in mz_list])
        flat = np.ascontiguousarray(np.concatenate(mz_list)) if n else np.zeros(0, dtype=m128)
        a, ao, rl, mp, mo = (C.c_void_p() for _ in range(5))
        _check(L.mga_seed_batch(self.gi, n, flat.ctypes.data, off.ctypes.data,
                                self.mo.occ_max1 if max_occ is None else max_occ,
                                C.byref(a), C.byref(ao), C.byref(rl), C.byref(mp), C.byref(mo)), "mga_seed_batch")
        aoff = _take(ao, n + 1, np.int64)
        moff = _take(mo, n + 1, np.int64)
        aa = _take(a, int(aoff[-1]), m128)
        rep = _take(rl, n, np.int32)
        mini = _take(mp, int(moff[-1]), np.int32)
        return [(aa[aoff[i]:aoff[i + 1]], int(rep[i]), mini[moff[i]:moff[i + 1]]) for i in range(n)]


def lchain_batch(anchor_list, **kw):
    """mg_lchain_dp for each read's x-sorted anchors -> list of (u, compacted anchors)"""
    L = load()
    n = len(anchor_list)
    par = lchain_par_t(kw.get("max_dist_x", 5000), kw.get("max_dist_y", 5000), kw.get("bw", 500),
                       kw.get("max_skip", 25), kw.get("max_iter", 5000), kw.get("min_cnt", 5), kw.get("min_sc", 40),
                       kw.get("pen_gap", 1.0), kw.get("pen_skip", 0.05))
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(a) for a in anchor_list])
    flat = np.ascontiguousarray(np.concatenate(anchor_list)) if n else np.zeros(0, dtype=m128)
    u, uo, b, bo = (C.c_void_p() for _ in range(4))
    _check(L.mga_lchain_batch(n, flat.ctypes.data, off.ctypes.data, C.byref(par), C.byref(u), C.byref(uo),
                              C.byref(b), C.byref(bo)), "mga_lchain_batch")
    uoff = _take(uo, n + 1, np.int64)
    boff = _take(bo, n + 1, np.int64)
    uu = _take(u, int(uoff[-1]), np.uint64)
    bb = _take(b, int(boff[-1]), m128)
    return [(uu[uoff[i]:uoff[i + 1]], bb[boff[i]:boff[i + 1]]) for i in range(n)]


def gaf_div_batch(div):
    """dv:f: text of each float as the device's GAF writer prints it -> list of bytes"""
    L = load()
    div = np.ascontiguousarray(div, dtype=np.float32)
    out = np.zeros(len(div) * 8, dtype=np.uint8)
    L.mga_gaf_div_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    _check(L.mga_gaf_div_batch(len(div), div.ctypes.data, out.ctypes.data), "mga_gaf_div_batch")
    return [bytes(out[8 * i:8 * i + 8]).rstrip(b"\0") for i in range(len(div))]


def sort128x_batch(arrays):
    """radix_sort_128x on the device, one wavefront per array (the chaining kernels' sort: LDS form up to 1024 elements) -> list of sorted copies"""
    L = load()
    n = len(arrays)
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(a) for a in arrays])
    flat = np.ascontiguousarray(np.concatenate(arrays)) if n else np.zeros(0, dtype=m128)
    L.mga_sort128x_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    _check(L.mga_sort128x_batch(n, flat.ctypes.data, off.ctypes.data), "mga_sort128x_batch")
    return [flat[off[i]:off[i + 1]] for i in range(n)]


def map_files(graph_path, read_paths, out_path, preset="lr", cigar=True, n_threads=8, verbose=1, idx_opt=None, map_opt=None, flags=0):
    """gfa_read + mg_map_files: the whole `minigraph -cx lr graph reads > out` job through the C ABI.
    idx_opt / map_opt: {field: value} written into mg_idxopt_t / mg_mapopt_t after the preset, the way main.c:131-191 applies
    command-line options; flags: MG_M_* bits OR-ed into mg_mapopt_t.flag."""
    L = load()
    io, mo, go = idxopt_t(), mapopt_t(), ggopt_t()
    L.mg_opt_set(None, C.byref(io), C.byref(mo), C.byref(go))
    if L.mg_opt_set(preset.encode(), C.byref(io), C.byref(mo), C.byref(go)) != 0:
        raise ValueError("unknown preset %r" % preset)
    if cigar:
        mo.flag |= MG_M_CIGAR
    mo.flag |= flags
    for k, v in (idx_opt or {}).items():
        setattr(io, k, v)
    for k, v in (map_opt or {}).items():
        setattr(mo, k, v)
    if L.mg_opt_check(C.byref(io), C.byref(mo), C.byref(go)) != 0:
        raise ValueError("mg_opt_check rejected the options")
    C.c_int.in_dll(L, "mg_verbose").value = verbose
    g = L.gfa_read(graph_path.encode())
    if not g:
        raise RuntimeError("gfa_read(%s) failed" % graph_path)
    fns = (C.c_char_p * len(read_paths))(*[p.encode() for p in read_paths])
    rc = L.mga_map_files_to_path(g, len(read_paths), fns, C.byref(io), C.byref(mo), n_threads, out_path.encode())
    L.gfa_destroy(g)
    if rc != 0:
        raise RuntimeError("mapping failed: %s" % L.mga_last_error().decode())


class MappedGaf:
    """GAF text of one mga_map_files_shard() call: a malloc'ed buffer owned by this object (zero-copy numpy view, bytes on request)"""

    def __init__(self, ptr, n, seg_len, t_map, cap=0):
        self.ptr, self.n, self.seg_len, self.t_map, self.cap = ptr, n, seg_len, t_map, cap

    def __len__(self):
        return self.n

    def view(self):
        return np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), shape=(self.n,)) if self.n else np.zeros(0, dtype=np.uint8)

    def bytes(self):
        return C.string_at(self.ptr, self.n) if self.n else b""

    def free(self):
        if self.ptr is not None and self.ptr.value:
            load().mga_free(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def map_files_idx(graph, read_paths, n_threads=8, rank=0, world=1, reuse=None, out_path=None):
    """the mapping phase of mg_map_files() against an existing index: FASTA/FASTQ files -> GAF text in memory (MappedGaf).
    world > 1: this process maps shard `rank` (see mga_map_files_shard in include/minigraph_amd.h); .seg_len lists its bytes per
    output segment, .t_map is the wall time of the phase (reader + pipeline + sink) measured inside the library.
    reuse: a MappedGaf of an earlier call whose buffer may be overwritten (it is consumed).
    out_path: the GAF goes to this FILE through the library's writer thread (the reference's step 2, gmap.c:119-139) instead of a buffer; returns the seconds of the phase."""
    L = load()
    fns = (C.c_char_p * len(read_paths))(*[p.encode() for p in read_paths])
    mem, n, cap, seg, nseg, t = C.c_void_p(), C.c_int64(0), C.c_int64(0), C.c_void_p(), C.c_int(0), C.c_double(0.0)
    if out_path is not None:
        libc = C.CDLL(None)
        libc.fopen.restype, libc.fopen.argtypes, libc.fclose.argtypes = C.c_void_p, [C.c_char_p, C.c_char_p], [C.c_void_p]
        fp = libc.fopen(out_path.encode(), b"wb")
        if not fp:
            raise OSError("cannot open %s" % out_path)
        try:
            _check(L.mga_map_files_shard(graph.gi, len(read_paths), fns, C.byref(graph.mo), n_threads, rank, world, C.c_void_p(fp),
                                         None, None, None, C.byref(seg), C.byref(nseg), C.byref(t)), "mga_map_files_shard")
        finally:
            libc.fclose(fp)
        L.mga_free(seg)
        return t.value
    if reuse is not None and reuse.ptr is not None and reuse.ptr.value and reuse.cap > 0:
        mem, cap = reuse.ptr, C.c_int64(reuse.cap)
        reuse.ptr, reuse.n = None, 0
    _check(L.mga_map_files_shard(graph.gi, len(read_paths), fns, C.byref(graph.mo), n_threads, rank, world, None,
                                 C.byref(mem), C.byref(n), C.byref(cap), C.byref(seg), C.byref(nseg), C.byref(t)), "mga_map_files_shard")
    seg_len = _take(seg, nseg.value, np.int64)
    return MappedGaf(mem, n.value, seg_len, t.value, cap.value)


def reads_shard_dump(path, out_path, rank, world, batch_bases=500000000, n_threads=4):
    """the reads of shard rank/world as the sharded reader cuts them -> one-line FASTA in out_path; returns records per segment"""
    L = load()
    seg, nseg = C.c_void_p(), C.c_int(0)
    _check(L.mga_reads_shard_dump(path.encode(), batch_bases, n_threads, rank, world, out_path.encode(), C.byref(seg), C.byref(nseg)),
           "mga_reads_shard_dump")
    return _take(seg, nseg.value, np.int64)


class kstring_t(C.Structure):
    _fields_ = [("l", C.c_uint), ("m", C.c_uint), ("s", C.c_void_p)]


def map_batch_api(graph, names, seqs, n_threads=4, per_read=False):
    """The reference-shaped C API, as a caller of minigraph.h would use it: mg_map_batch() (or mg_map() read by read when
    per_read=True) -> mg_gchains_t objects -> mg_write_gaf() per read -> mg_gchain_free().  Returns the GAF bytes."""
    L = load()
    n = len(seqs)
    L.mg_map.restype = C.c_void_p
    L.mg_tbuf_init.restype = C.c_void_p
    L.mg_map.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.POINTER(mapopt_t), C.c_char_p]
    L.mg_map_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_void_p),
                               C.POINTER(mapopt_t), C.c_int]
    L.mg_write_gaf.argtypes = [C.POINTER(kstring_t), C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_char_p, C.c_uint64, C.c_void_p]
    L.mg_write_gaf.restype = None
    L.mg_gchain_free.argtypes = [C.c_void_p]
    L.mg_gchain_free.restype = None
    L.mg_tbuf_destroy.argtypes = [C.c_void_p]
    gcs = (C.c_void_p * n)()
    if per_read:
        tb = L.mg_tbuf_init()
        for i in range(n):
            gcs[i] = L.mg_map(graph.gi, len(seqs[i]), seqs[i], tb, C.byref(graph.mo), names[i])
        L.mg_tbuf_destroy(tb)
    else:
        qlens = (C.c_int * n)(*[len(s) for s in seqs])
        sp, npp = (C.c_char_p * n)(*seqs), (C.c_char_p * n)(*names)
        _check(L.mg_map_batch(graph.gi, n, qlens, sp, npp, gcs, C.byref(graph.mo), n_threads), "mg_map_batch")
    out, ks = [], kstring_t(0, 0, None)
    for i in range(n):
        ql = C.c_int32(len(seqs[i]))
        L.mg_write_gaf(C.byref(ks), graph.g, gcs[i], 1, C.byref(ql), names[i], graph.mo.flag, None)
        if ks.l:
            out.append(C.string_at(ks.s, ks.l))
        L.mg_gchain_free(gcs[i])
    L.mga_free(ks.s)
    return b"".join(out)


class Reads:
    """mga_reads_load(): a read set resident in host memory and HBM"""

    def __init__(self, path, max_reads=0):
        L = load()
        self.h = L.mga_reads_load(path.encode(), max_reads)
        if not self.h:
            raise RuntimeError("mga_reads_load failed: %s" % L.mga_last_error().decode())
        self.n, self.bases = L.mga_reads_count(self.h), L.mga_reads_bases(self.h)

    def close(self):
        if self.h:
            load().mga_reads_free(self.h)
        self.h = None


class GafBuffer:
    """the GAF text of one pass, owned by the index inside the C library: valid until the next map_reads() on the same
    Graph (no copy until bytes() is asked for)"""

    def __init__(self, ptr, n):
        self.ptr, self.n = ptr, n

    def __len__(self):
        return self.n

    def view(self):
        return np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), shape=(self.n,)) if self.n else np.zeros(0, dtype=np.uint8)

    def bytes(self):
        return C.string_at(self.ptr, self.n)

    def free(self):
        self.ptr = None


def map_reads(graph, reads, n_threads=8, copy=True):
    """one pass of the hot path over a resident read set -> GAF bytes (or a GafBuffer when copy=False)"""
    L = load()
    buf, n = C.c_void_p(), C.c_int64(0)
    _check(L.mga_map_reads(graph.gi, reads.h, C.byref(graph.mo), n_threads, C.byref(buf), C.byref(n)), "mga_map_reads")
    gb = GafBuffer(buf, n.value)
    if not copy:
        return gb
    out = gb.bytes()
    gb.free()
    return out


def get_stats(graph, reset=False):
    st = stats_t()
    load().mga_get_stats(graph.gi, C.byref(st), 1 if reset else 0)
    return {k: getattr(st, k) for k, _ in st._fields_}


def prof_enable(on=True):
    load().mga_prof_enable(1 if on else 0)


def prof_get(reset=False):
    ms = np.zeros(len(KERNELS), dtype=np.float64)
    cnt = np.zeros(len(KERNELS), dtype=np.int64)
    load().mga_prof_get(ms.ctypes.data, cnt.ctypes.data, 1 if reset else 0)
    return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(KERNELS)}
