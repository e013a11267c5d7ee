"""Profiling aid (GPU box): where part 2 of graph chaining on the device (k_gchain_p2, a wavefront per bridge) spends its time.
MGA_GC_SPLIT_DEBUG=2 prints per chunk the distribution of bridge durations and runs the longest bridge once more alone with per-stage cycle sums.
  python minigraph_amd/tools/gc_job_prof.py [genome] [reads]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import minigraph_amd as mga
genome = sys.argv[1] if len(sys.argv) > 1 else "400000000"
reads = sys.argv[2] if len(sys.argv) > 2 else "16384"
d = tempfile.mkdtemp()
subprocess.check_call([mga.MGSIM, "-p", d + "/w", "-G", genome, "-c", "8", "-H", "5", "-n", reads, "-s", "11"], stderr=subprocess.DEVNULL)
child = """
import sys; sys.path.insert(0, %r); import minigraph_amd as mga
G = mga.Graph(sys.argv[1], n_threads=16); m = mga.map_files_idx(G, [sys.argv[2]], n_threads=16); m.free()
""" % ROOT
p = subprocess.run([sys.executable, "-c", child, d + "/w.gfa", d + "/w.reads.fa"], env=dict(os.environ, MGA_DEV_GCHAIN="1", MGA_PIPE="1", MGA_GC_SPLIT_DEBUG="2"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
print("\n".join(l for l in p.stderr.decode().splitlines() if "part 2" in l or "longest" in l or "rror" in l)[-6000:])
