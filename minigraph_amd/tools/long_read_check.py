"""ultra-long -x lr reads: 2 x 5 Mbp reads, the whole job file -> file here and in the reference; same bytes, and the times.
  python minigraph_amd/tools/long_read_check.py [read_len] [n]"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import minigraph_amd as mga
L, n = (sys.argv[1] if len(sys.argv) > 1 else "5000000"), (sys.argv[2] if len(sys.argv) > 2 else "2")
d = tempfile.mkdtemp()
subprocess.check_call([mga.MGSIM, "-p", d + "/t", "-G", "16000000", "-H", "3", "-n", n, "-l", L, "-e", "0.05", "-s", "38"], stderr=subprocess.DEVNULL)
g, r = d + "/t.gfa", d + "/t.reads.fa"
out = {}
for tag, env in (("long_read_placement", {}), ("one_wavefront_per_read", {"MGA_LONG_READ": "0"})):
    e = dict(os.environ, **env)
    code = "import sys,time; sys.path.insert(0,%r); import minigraph_amd as mga; t0=time.time(); mga.map_files(sys.argv[1],[sys.argv[2]],sys.argv[3],preset='lr',cigar=True,n_threads=16,verbose=0); print('T',time.time()-t0)" % ROOT
    p = subprocess.run([sys.executable, "-c", code, g, r, d + "/" + tag + ".gaf"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out[tag + "_s"] = round(float([l for l in p.stdout.decode().splitlines() if l.startswith("T")][0].split()[1]), 2) if p.returncode == 0 else p.stderr.decode()[-300:]
t0 = time.time()
with open(d + "/ref.gaf", "wb") as fo:
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "minigraph"), "-c", "-x", "lr", "-t", "16", g, r], stdout=fo, stderr=subprocess.DEVNULL)
out["reference_s"] = round(time.time() - t0, 2)
out["parity"] = [subprocess.call(["cmp", "-s", d + "/" + t + ".gaf", d + "/ref.gaf"]) == 0 for t in ("long_read_placement", "one_wavefront_per_read")]
out["workload"] = "%s x %s bp reads (5%% errors) vs a 16 Mbp 3-haplotype graph, -cx lr, file -> file incl. graph load + index on both sides" % (n, L)
print(json.dumps(out))
