"""Profiling aid (GPU box): per-stage cycles of the HOST instantiation of graph chaining (gc_core.h built with -DGC_HOST_PROF: `python minigraph_amd/tools/gchain_ab.py`-style
variant lib_hprof.so) on a bench-like workload.   python minigraph_amd/tools/host_gc_prof.py [genome] [reads]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MGA_LIB", os.path.join(ROOT, "minigraph_amd", "lib", "ab", "lib_hprof.so"))
os.environ["MGA_DEV_GCHAIN"] = "0"
import minigraph_amd as mga
genome = sys.argv[1] if len(sys.argv) > 1 else "1000000000"
reads = sys.argv[2] if len(sys.argv) > 2 else "40000"
d = tempfile.mkdtemp()
subprocess.check_call([mga.MGSIM, "-p", d + "/w", "-G", genome, "-c", "8", "-H", "5", "-n", reads, "-s", "11"], stderr=subprocess.DEVNULL)
G = mga.Graph(d + "/w.gfa", n_threads=16)
for _ in range(2):
    m = mga.map_files_idx(G, [d + "/w.reads.fa"], n_threads=16); m.free()
L = mga.load()
L.mga_gc_host_prof_dump()
