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
