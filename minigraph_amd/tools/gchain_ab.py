"""A/B timing of k_gchain builds on one GPU box (a profiling aid).

  python minigraph_amd/tools/gchain_ab.py build   # here: variants of libminigraph_amd.so that differ in k_gchain.hip's -D switches
  python minigraph_amd/tools/gchain_ab.py run     # on the GPU box: one workload, every variant in its own process, k_gchain ms per variant

Variants live in minigraph_amd/lib/ab/ (git-ignored like every built file; they travel with gpurun)."""
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "minigraph_amd", "csrc")
LIB = os.path.join(ROOT, "minigraph_amd", "lib")
AB = os.path.join(LIB, "ab")
HIPCC = "/opt/rocm/bin/hipcc"
VARIANTS = {
    "cur": [],                               # the tree as it is
    "nolds": ["-DGC_AB_NO_LDS_STATE"],       # round-2 placement of the routines' state (a private copy per lane: scratch memory)
    "ni": ["-DGC_AB_NOINLINE"],              # the big routines as functions of their own (half the code)
    "w4": ["-DGC_AB_WAVES_PER_EU=4"],        # 128 VGPRs, four resident wavefronts per SIMD
    "mono": None,                            # (round 4) the tree's own library, one-kernel form (MGA_GC_SPLIT=0)
    "split": None,                           # the tree's own library, three-kernel form (the default)
    "p2w4": ["-DGC_AB_P2_WAVES=4"],          # part 2 (a wavefront per bridge) at 128 VGPRs, 4096 wavefronts
    "p2w3": ["-DGC_AB_P2_WAVES=3"],
}
ENV = {"w4": {"MGA_GC_WAVES": "4096"}, "mono": {"MGA_GC_SPLIT": "0"}, "split": {"MGA_GC_SPLIT": "1"}, "p2w4": {"MGA_GC_WAVES2": "4096"}, "p2w3": {"MGA_GC_WAVES2": "3072"}}


def build(names):
    os.makedirs(AB, exist_ok=True)
    others = [o for o in sorted(glob.glob(os.path.join(LIB, "obj", "*.o"))) if os.path.basename(o) != "k_gchain.hip.o"]
    for name in names:
        if VARIANTS[name] is None:
            continue
        obj = os.path.join(AB, name + ".o")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "-w"] + VARIANTS[name] +
                              ["-c", os.path.join(CSRC, "k_gchain.hip"), "-o", obj])
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic"] + others + [obj, "-o", os.path.join(AB, "lib_%s.so" % name), "-lz", "-lpthread", "-lm"])
        os.remove(obj)
        print("built", name, VARIANTS[name])


CHILD = r"""
import os, sys, time, json
sys.path.insert(0, %(root)r)
import minigraph_amd as mga
import torch
os.environ["MGA_DEV_GCHAIN"] = "1"; os.environ["MGA_PIPE"] = "1"; os.environ["MGA_WFA_SIDE"] = "0"
G = mga.Graph(sys.argv[1], n_threads=16)
m = mga.map_files_idx(G, [sys.argv[2]], n_threads=16); ref = bytes(m.bytes()[:1 << 20]); n0 = len(m.bytes()); m.free()
mga.prof_enable(True); mga.prof_get(reset=True)
t0 = time.perf_counter()
for _ in range(2):
    m = mga.map_files_idx(G, [sys.argv[2]], n_threads=16); m.free()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 2
pr = mga.prof_get()
import hashlib
print("AB", json.dumps(dict(k_gchain_ms=round((pr["k_gchain"][0] + pr["k_gchain_p2"][0] + pr["k_gchain_p3"][0]) / 2, 2), parts_ms=[round(pr[k][0] / 2, 2) for k in ("k_gchain", "k_gchain_p2", "k_gchain_p3")], pass_ms=round(dt * 1e3, 1), gaf_bytes=n0, md5=hashlib.md5(ref).hexdigest())))
"""


def run(names, genome, reads):
    sys.path.insert(0, ROOT)
    import minigraph_amd as mga
    d = tempfile.mkdtemp(prefix="mga_ab_")
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "w"), "-G", str(genome), "-c", "8", "-H", "5", "-n", str(reads), "-s", "11"], stderr=subprocess.DEVNULL)
    import re
    for rep in ((1,) if os.environ.get("AB_PLAIN_ONLY") else (0, 1)):   # first round with per-stage cycle sums (MGA_GC_PROF=1: inflates the kernel, the sums compare builds), second round plain
        for name in names:
            lib = os.path.join(AB, "lib_%s.so" % name) if VARIANTS[name] is not None else os.path.join(LIB, "libminigraph_amd.so")
            if not os.path.exists(lib):
                continue
            env = dict(os.environ, MGA_LIB=lib, **ENV.get(name, {}))
            if rep == 0:
                env["MGA_GC_PROF"] = "1"
            try:
                p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, os.path.join(d, "w.gfa"), os.path.join(d, "w.reads.fa")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=int(os.environ.get("AB_TIMEOUT", "150")))
            except subprocess.TimeoutExpired:
                print(name, "TIMEOUT", flush=True)
                continue
            line = [l for l in p.stdout.decode().splitlines() if l.startswith("AB")]
            print(name, "prof" if rep == 0 else "plain", line[0] if line else ("FAILED " + p.stderr.decode()[-400:]), flush=True)
            if rep == 0:
                tot = {}
                for l in p.stderr.decode().splitlines():
                    if "gc-prof" in l:
                        for k, v in re.findall(r" ([A-Za-z:+()\-]+) ([0-9.]+)", l.split("Mcycles:")[1]):
                            tot[k] = max(tot.get(k, 0.0), float(v)) if k.startswith("MAX") else tot.get(k, 0.0) + float(v)
                print("   Gcycles over the 3 passes:", " ".join("%s %.1f" % (k, v * 1e-3) for k, v in tot.items()), "| SUM %.1f" % (sum(v for k, v in tot.items() if not k.startswith("MAX")) * 1e-3), flush=True)


def sq(name, genome, reads):
    """two SQ counter passes (rocprofv3 --pmc, no tracing) of one variant's child process; the k_gchain / k_plan rows of the summary"""
    sys.path.insert(0, ROOT)
    import minigraph_amd as mga
    d = tempfile.mkdtemp(prefix="mga_ab_")
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "w"), "-G", str(genome), "-c", "8", "-H", "5", "-n", str(reads), "-s", "11"], stderr=subprocess.DEVNULL)
    child = os.path.join(d, "child.py")
    open(child, "w").write(CHILD % {"root": ROOT})
    env = dict(os.environ, MGA_LIB=os.path.join(AB, "lib_%s.so" % name), TMPDIR="/tmp")
    if os.environ.get("AB_PMC") == "mem":
        sets_override = ["FETCH_SIZE", "WRITE_SIZE", "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum", "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"]
    else:
        sets_override = None
    sets = ["SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY",
            "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU",
            "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_SMEM"]
    dbs = []
    if sets_override:
        sets = sets_override
    for i, cs in enumerate(sets):
        out = os.path.join(d, "sq%d" % i)
        p = subprocess.run(["rocprofv3", "--pmc"] + cs.split() + ["-d", out, "-o", "pmc", "--", sys.executable, child, os.path.join(d, "w.gfa"), os.path.join(d, "w.reads.fa")],
                           env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        found = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
        print("pass", i, "rc", p.returncode, found[:1], flush=True)
        if not found:
            print(p.stderr.decode()[-600:])
        dbs += found[:1]
    import sqlite3
    acc = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        for kn, cn, calls, tot in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
            acc.setdefault(kn.split("(")[0][:28], {})[cn] = (float(tot), int(calls))
    for kn, k in sorted(acc.items()):
        if not any(x in kn for x in ("k_gchain", "k_plan", "k_lchain")):
            continue
        if sets_override:
            print(kn)
            for cn, (v, nl) in sorted(k.items()):
                print("    %-32s total %18.0f   per launch %16.1f (%d launches)" % (cn, v, v / max(1, nl), nl))
            continue
        w, cyc = k.get("SQ_WAVES", (1, 0))[0], k.get("SQ_WAVE_CYCLES", (1, 0))[0]
        print(kn, "launches", k.get("SQ_WAVES", (0, 0))[1], "waves %.0f" % w, "cycles/wave %.0f" % (cyc / w))
        for cn, (v, _) in sorted(k.items()):
            print("    %-24s per wave %14.1f   share of wave cycles %.4f" % (cn, v / w, v / cyc))


if __name__ == "__main__":
    if sys.argv[1] == "sq":
        sq(sys.argv[2], int(os.environ.get("AB_GENOME", "800000000")), int(os.environ.get("AB_READS", "49152")))
        sys.exit(0)
    names = [a for a in sys.argv[2:] if a in VARIANTS] or list(VARIANTS)
    if sys.argv[1] == "build":
        build(names)
    else:
        run(names, int(os.environ.get("AB_GENOME", "800000000")), int(os.environ.get("AB_READS", "49152")))
