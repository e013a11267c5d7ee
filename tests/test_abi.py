"""CPU-only: the C-ABI library loads and exports every function include/minigraph_amd.h declares."""
import ctypes
import os
import re

import minigraph_amd as mga

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "minigraph_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set()
    for m in re.finditer(r"^[A-Za-z_][\w \t\*]*?\b(\w+)\s*\([^;{]*\)\s*;", txt, flags=re.M):
        if not m.group(0).lstrip().startswith(("typedef", "#")):
            names.add(m.group(1))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(mga.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 25, names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    for var in ("mg_verbose", "mg_dbg_flag", "mg_realtime0"):
        assert hasattr(lib, var)


def test_struct_sizes_match_reference_abi():
    """sizes measured on the reference: mg_mapopt_t 168, mg_idxopt_t 12"""
    assert ctypes.sizeof(mga.mapopt_t) == 168
    assert ctypes.sizeof(mga.idxopt_t) == 12
    L = mga.load()
    io, mo, go = mga.idxopt_t(), mga.mapopt_t(), mga.ggopt_t()
    assert L.mg_opt_set(None, ctypes.byref(io), ctypes.byref(mo), ctypes.byref(go)) == 0
    assert (io.k, io.w, io.bucket_bits) == (17, 11, 14)
    assert (mo.occ_max1, mo.bw, mo.bw_long, mo.max_gap, mo.min_lc_cnt, mo.min_gc_score) == (50, 500, 20000, 5000, 5, 50)
    assert L.mg_opt_set(b"asm", ctypes.byref(io), ctypes.byref(mo), ctypes.byref(go)) == 0
    assert (io.k, io.w, mo.bw, mo.bw_long) == (19, 10, 1000, 150000) and (mo.flag & 0x8000)
    assert L.mg_opt_set(b"nope", ctypes.byref(io), ctypes.byref(mo), ctypes.byref(go)) == -1


def test_struct_layouts_match_reference_headers():
    """every struct of the boundary: sizeof and every field offset of include/minigraph_amd.h, measured by a compiled C
    probe, against the same probe compiled with the reference's minigraph.h / gfa.h / mgpriv.h (tests/golden/abi_layout.json, made by
    tests/golden/make_abi_layout.py; re-derived here when /root/reference is present so that the committed file cannot go stale)"""
    import json
    import abi_probe
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "abi_layout.json")))
    ours = abi_probe.run_probe(["-I" + os.path.join(ROOT, "include")], ['#include "minigraph_amd.h"'])
    assert ours == gold, {k: (gold.get(k), ours.get(k)) for k in set(gold) | set(ours) if gold.get(k) != ours.get(k)}
    for k, v in (("mg128_t", 16), ("mg_idxopt_t", 12), ("mg_mapopt_t", 168), ("mg_idx_t", 48), ("mg_lchain_t", 40), ("mg_llchain_t", 20),
                 ("mg_cigar_t", 24), ("mg_ds_t", 24), ("mg_gchain_t", 104), ("mg_gchain_t.p", 72), ("mg_gchain_t.ds", 80),
                 ("mg_gchains_t", 48), ("gfa_arc_t", 32), ("gfa_seg_t", 64), ("gfa_edseq_t", 16)):
        assert gold[k] == v, (k, gold[k], v)   # the numbers SURVEY 8b quotes
    if os.path.exists("/root/reference/minigraph.h"):
        ref = abi_probe.run_probe(["-I/root/reference"], ['#include "minigraph.h"', '#include "gfa.h"', '#include "mgpriv.h"'])
        assert ref == gold


def test_no_gpu_fails_loudly():
    """without a GPU every compute entry point must fail with an error, never fall back to the CPU"""
    L = mga.load()
    if L.mga_device_count() > 0:
        return
    try:
        mga.sketch_batch([b"ACGTACGTACGTACGTACGTACGTACGT"], 5, 4)
    except RuntimeError as e:
        assert "no HIP device" in str(e) or "failed" in str(e)
    else:
        raise AssertionError("sketch_batch succeeded without a GPU")


def test_graph_image_loader_rejects_truncated_and_corrupt_files(tmp_path):
    """the image loader maps a file and used to trust every offset / count in it.  Every section must lie inside the file, the sequence offsets must
    match the segment lengths, names must be terminated, arcs must name vertices of the graph.  (The checks run before the device is touched: CPU test.)"""
    import struct
    import ctypes as C
    import numpy as np
    L = mga.load()
    gfa = tmp_path / "g.gfa"
    rng = np.random.default_rng(2)
    with open(gfa, "w") as f:
        for s in range(3):
            f.write("S\ts%d\t%s\tSN:Z:chr1\tSO:i:%d\tSR:i:0\n" % (s, "".join(rng.choice(list("ACGT"), 500 + 100 * s)), 1000 * s))
        f.write("L\ts0\t+\ts1\t+\t0M\nL\ts1\t+\ts2\t+\t0M\n")
    L.gfa_read.restype = C.c_void_p
    L.gfa_read.argtypes = [C.c_char_p]
    g = L.gfa_read(str(gfa).encode())
    assert g
    L.mga_graph_image_save.argtypes = [C.c_void_p, C.c_char_p]
    img = tmp_path / "g.mgi"
    assert L.mga_graph_image_save(g, str(img).encode()) == 0
    good = bytearray(open(img, "rb").read())
    L.mga_index_load_image.restype = C.c_void_p
    L.mga_index_load_image.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p]
    L.mga_last_error.restype = C.c_char_p
    io, mo, go = mga.idxopt_t(), mga.mapopt_t(), mga.ggopt_t()
    L.mg_opt_set(None, C.byref(io), C.byref(mo), C.byref(go))
    # header: magic 8, version 4, pad 4, then 5 counts (n_seg n_sseq n_arc max_rank tot_seq) and 11 offsets / sizes, all uint64
    names = ["n_seg", "n_sseq", "n_arc", "max_rank", "tot_seq", "off_seg", "off_names", "names_bytes", "off_sseq", "off_snames", "snames_bytes", "off_arc", "off_idx",
             "off_seqoff", "off_seq", "file_bytes"]
    hdr = dict(zip(names, struct.unpack_from("<16Q", good, 16)))

    def attempt(buf, label):
        p = tmp_path / ("bad_%s.mgi" % label)
        open(p, "wb").write(bytes(buf))
        gi = L.mga_index_load_image(str(p).encode(), C.byref(io), 2, C.byref(mo))
        msg = L.mga_last_error().decode()
        assert not gi, label
        return msg

    def patched(field, value):
        b = bytearray(good)
        struct.pack_into("<Q", b, 16 + 8 * names.index(field), value)
        return b

    assert "not a graph image" in attempt(b"\0" * 4096, "zeros")
    for label, buf in [
        ("truncated", good[:len(good) // 2]),
        ("n_seg_huge", patched("n_seg", 1 << 40)),
        ("n_arc_wrap", patched("n_arc", (1 << 64) // 32 + 1)),           # count * size wraps in 64 bits
        ("off_seq_wrap", patched("off_seq", (1 << 64) - 8)),             # off_seq + tot_seq wraps
        ("off_arc_out", patched("off_arc", len(good) + 64)),
        ("off_idx_out", patched("off_idx", len(good) - 8)),
        ("tot_seq_short", patched("tot_seq", hdr["tot_seq"] - 1)),       # offsets no longer end at tot_seq
        ("names_short", patched("names_bytes", hdr["names_bytes"] - 1)), # last name not terminated
    ]:
        msg = attempt(buf, label)
        assert "truncated or corrupt" in msg, (label, msg)
    b = bytearray(good)   # a segment record whose length disagrees with the sequence offsets
    struct.pack_into("<i", b, hdr["off_seg"], 1 << 30)
    assert "truncated or corrupt" in attempt(b, "seg_len")
    b = bytearray(good)   # a name offset outside the name block
    struct.pack_into("<I", b, hdr["off_seg"] + 20, 1 << 20)
    assert "truncated or corrupt" in attempt(b, "name_off")
    b = bytearray(good)   # an arc to a vertex the graph does not have
    struct.pack_into("<I", b, hdr["off_arc"] + 8, 1000)
    assert "truncated or corrupt" in attempt(b, "arc_w")
    b = bytearray(good)   # an arc-index entry that runs past the arc array
    struct.pack_into("<Q", b, hdr["off_idx"], (0 << 32) | 1000)
    assert "truncated or corrupt" in attempt(b, "idx_cnt")
    # the untouched image passes the file checks: on a box without a GPU the failure is the device's
    gi = L.mga_index_load_image(str(img).encode(), C.byref(io), 2, C.byref(mo))
    if not gi:
        assert "corrupt" not in L.mga_last_error().decode() and "not a graph image" not in L.mga_last_error().decode()
    else:
        L.mg_idx_destroy.argtypes = [C.c_void_p]
        L.mg_idx_destroy(gi)
