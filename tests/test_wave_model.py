"""CPU: the wavefront-resident blocks of graph chaining's GWFA (gc_core.h: gc_gw_extend_block, gc_gw_dedup_block, gc_gw_dedup_wave,
gc_intv_add_wave) are device-only code paths -- a wavefront's 64 lanes hold a GWFA wavefront's cells and exchange them by lane
shuffles, ballots and ds_permute.  They are written in a wave-vector notation that ALSO compiles as a 64-lane MODEL on the host
(-DGC_WAVE_MODEL: vectors become arrays of 64, the per-lane body loops over the lanes).  This test builds the library with its host
instantiation switched to that model and runs the host parity tests -- the reference's own gfa_ed_step, and whole jobs against the
reference binary -- through it: the source the device runs, checked on the CPU.  (The default host build keeps the sequential code.)"""
import ctypes as C
import glob
import os
import subprocess
import sys
import tempfile

import pytest

import minigraph_amd as mga
import refbind as rb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "minigraph_amd", "csrc")
OBJ = os.path.join(ROOT, "minigraph_amd", "lib", "obj")
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(mga.LIB_PATH) and rb.have_oracle()), reason="hipcc / libraries not available")


@pytest.fixture(scope="module")
def model_lib():
    d = tempfile.mkdtemp(prefix="mga_wavemodel_")
    obj = os.path.join(d, "k_gchain.hip.o")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "-w",
                           "-DGC_WAVE_MODEL", "-DGC_STATS", "-c", os.path.join(CSRC, "k_gchain.hip"), "-o", obj])
    others = [o for o in sorted(glob.glob(os.path.join(OBJ, "*.o"))) if os.path.basename(o) != "k_gchain.hip.o"]
    lib = os.path.join(d, "libminigraph_amd.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic"] + others + [obj, "-o", lib, "-lz", "-lpthread", "-lm"])
    return lib


def test_host_parity_suites_through_the_wave_model(model_lib):
    env = dict(os.environ, MGA_LIB=model_lib)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_host_gwfa.py"),
                        os.path.join(ROOT, "tests", "test_host_pipeline.py")], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-3000:]
    assert " passed" in out and "skipped" not in out.splitlines()[-1], out[-500:]


CHILD = r"""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import minigraph_amd as mga
import hostpipe as hp
d = sys.argv[1]
subprocess.check_call([mga.MGSIM, "-p", d + "/t", "-G", "12000000", "-c", "3", "-H", "5", "-n", "250", "-s", "23"], stderr=subprocess.DEVNULL)
want, occ, lco = hp.run_reference(d + "/t.gfa", d + "/t.reads.fa", cigar=False)
got, _ = hp.map_with_oracle_stages(d + "/t.gfa", d + "/t.reads.fa", occ, lco, cigar=False, n_threads=1)
out = (C.c_longlong * 8)()
mga.load().mga_gc_stats_get(out)
print("RESULT", int(got == want), len(got), *list(out))
"""


@pytest.mark.skipif(not os.path.exists(rb.REF_BIN), reason="oracle/_ref/minigraph not built")
def test_bench_shaped_reads_through_the_wave_model_and_the_paths_it_took(model_lib):
    """250 x 10 kb reads at 10 % error on a 5-haplotype bubble graph (the bench workload's shape) vs the reference binary, and proof that the
    wave paths did the work: block extensions, register dedups, two-pass dedups and interval merges all ran, none handed a wavefront back"""
    d = tempfile.mkdtemp(prefix="mga_wavemodel_job_")
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, d], env=dict(os.environ, MGA_LIB=model_lib), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    res = [l for l in p.stdout.decode().splitlines() if l.startswith("RESULT")][0].split()[1:]
    same, n_bytes, calls, steps, blk_ext, dedups, blk_dedup, blk_wave, blk_fallback, blk_merge = map(int, res)
    assert same == 1 and n_bytes > 10000
    assert calls > 100 and steps > 1000
    assert blk_ext >= steps           # every step went through gc_gw_extend_block at least once
    assert blk_dedup > 100 and blk_wave > 100 and blk_merge > 100
    assert blk_dedup + blk_wave > 0.9 * dedups
    assert blk_fallback == 0
