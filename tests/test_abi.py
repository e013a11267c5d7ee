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
