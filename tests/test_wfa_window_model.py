"""CPU: the exactness claim behind the windowed WFA tiers (minigraph_amd/csrc/k_wfa_w.hip).  The oracle's exact WFA (oracle/mgo_wfa.c, pinned against the
reference's mwf_wfa_exact in test_oracle_vs_ref.py) is restricted to a window of diagonals by a three-line patch applied here -- cells outside the window read
NEG_INF, the run stops when the score reaches the window's bound -- and must return the unrestricted score and CIGAR whenever it returns at all."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_windowed_wfa_equals_the_full_band_below_the_bound():
    src = open(os.path.join(ROOT, "oracle", "mgo_wfa.c")).read()
    sig = ("int32_t mgo_wfa_exact(const mgo_wfa_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs,\n"
           "\t\t\t\t\t  uint32_t *cigar, int32_t cap, int32_t *n_cigar, int64_t *n_iter_)")
    band = "\t\tlo = wlo > -tl ? wlo - 1 : -tl;\n\t\thi = whi < ql ? whi + 1 : ql;\n\t\t++s;"
    assert sig in src and band in src, "oracle/mgo_wfa.c changed: update the patch of this test"
    win = src.replace(sig, sig.replace("mgo_wfa_exact(", "mgo_wfa_win(").replace("int64_t *n_iter_)", "int64_t *n_iter_, int32_t WL, int32_t WR, int32_t WB)"))
    win = win.replace(band, "\t\tlo = wlo > -tl ? wlo - 1 : -tl;\n\t\thi = whi < ql ? whi + 1 : ql;\n\t\tif (lo < WL) lo = WL;\n\t\tif (hi > WR) hi = WR;\n"
                            "\t\tif (lo > hi || s + 1 >= WB) { stopped = 1; break; } /* (lo > hi: the window holds no reachable diagonal, e.g. not even diagonal 0) */\n\t\t++s;")
    d = tempfile.mkdtemp()
    open(os.path.join(d, "wfa_win.c"), "w").write(win)
    exe = os.path.join(d, "check")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "cmodels", "wfa_window_check.c"),
                           os.path.join(d, "wfa_win.c"), os.path.join(ROOT, "oracle", "mgo_wfa.c"), "-o", exe])
    p = subprocess.run([exe, "12000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-1500:])
    assert b"mismatches 0" in p.stdout, p.stdout


def test_packed_two_diagonals_per_lane_kernel_model_equals_the_oracle():
    """the PACKED forward pass of the 128 / 192 / 256-diagonal rungs (k_wfa_fwp: two neighbouring diagonals in the 16-bit halves of a lane's register, the recurrence on packed
    16-bit arithmetic with wrap-around differences as tie-break masks, traceback rows holding reachable diagonals only) restated lane by lane in C
    (tests/cmodels/wfa_packed_model.c) + the walk of k_wfa_tb: score and CIGAR are the oracle's whenever the window decides, the model stops within its bound and the walk
    never reads a traceback dword the forward pass did not write"""
    d = tempfile.mkdtemp()
    exe = os.path.join(d, "model")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "cmodels", "wfa_packed_model.c"),
                           os.path.join(ROOT, "oracle", "mgo_wfa.c"), "-o", exe])
    p = subprocess.run([exe, "4000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-1500:])
    assert b"mismatches 0" in p.stdout, p.stdout
