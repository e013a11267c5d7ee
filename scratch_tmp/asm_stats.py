import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import minigraph_amd as mga
mga.load()
d = tempfile.mkdtemp()
L = int(sys.argv[1]); n = int(sys.argv[2]); cigar = len(sys.argv) > 3 and sys.argv[3] == "c"
subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "a"), "-G", "50000000", "-H", "3", "-n", str(n), "-l", str(L), "-e", "0.001", "-s", "5"], stderr=subprocess.DEVNULL)
t0 = time.time(); G = mga.Graph(os.path.join(d, "a.gfa"), preset="asm", cigar=cigar, n_threads=16); t1 = time.time()
R = mga.Reads(os.path.join(d, "a.reads.fa")); t2 = time.time()
mga.prof_enable(True)
for rep in range(2):
    mga.get_stats(G, reset=True); mga.prof_get(reset=True)
    t2 = time.time(); out = mga.map_reads(G, R, n_threads=16, copy=False); t3 = time.time()
    st = mga.get_stats(G)
    print("contig", L, "n", n, "cigar", cigar, "index %.2f map %.2f" % (t1 - t0, t3 - t2), flush=True)
    print({k: round(st[k], 3) for k in ("t_sketch", "t_seed", "t_lchain", "t_host_chain", "t_wfa", "t_host_post", "t_gaf")}, "n_mz", st["n_mz"], "n_hit", st["n_hit"], "n_wfa", st["n_wfa"], "cells", st["wfa_cells"])
    print({k: (round(v[0], 1), v[1]) for k, v in mga.prof_get().items() if v[0] > 0.05})
