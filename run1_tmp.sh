set -e
cd /root/repo
python - <<'PY'
import os, subprocess, tempfile, sys, time
sys.path.insert(0, "tests")
import minigraph_amd as mga
import refbind as rb
def run(tag, simargs, preset="lr", cigar=True, n=None):
    d = tempfile.mkdtemp()
    subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t")] + simargs, stderr=subprocess.DEVNULL)
    graph, reads = os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")
    ref = os.path.join(d, "ref.gaf")
    t0 = time.time()
    with open(ref, "wb") as fo:
        subprocess.check_call([rb.REF_BIN] + (["-c"] if cigar else []) + ["-x", preset, "-t", "16", graph, reads], stdout=fo, stderr=subprocess.DEVNULL)
    t1 = time.time()
    G = mga.Graph(graph, preset=preset, cigar=cigar, n_threads=16)
    R = mga.Reads(reads)
    mga.get_stats(G, reset=True)
    got = mga.map_reads(G, R, n_threads=16)
    t2 = time.time()
    st = mga.get_stats(G)
    want = open(ref, "rb").read()
    print("%-28s %s  ref %.1fs ours %.1fs  lines %d  rescue dev/host %d/%d" % (tag, "IDENTICAL" if got == want else "MISMATCH", t1 - t0, t2 - t1, got.count(b"\n"), st["n_rescue_dev"], st["n_rescue_host"]), flush=True)
    if got != want:
        a, b = want.split(b"\n"), got.split(b"\n")
        for i, (x, y) in enumerate(zip(a, b)):
            if x != y:
                fx, fy = x.split(b"\t"), y.split(b"\t")
                for k, (p, q) in enumerate(zip(fx, fy)):
                    if p != q:
                        print("  line %d field %d\n   ref %r\n   got %r" % (i, k, p[:200], q[:200])); break
                break
    R.close(); G.close()
run("50kb reads", ["-G", "5000000", "-H", "3", "-n", "300", "-l", "50000", "-s", "31"])
run("100kb reads err 0.15", ["-G", "5000000", "-H", "3", "-n", "100", "-l", "100000", "-e", "0.15", "-s", "32"])
run("1kb reads", ["-G", "3000000", "-H", "3", "-n", "5000", "-l", "1000", "-s", "33"])
run("err 0.2", ["-G", "3000000", "-H", "3", "-n", "1500", "-e", "0.2", "-s", "34"])
run("5 haplotypes, 3 chr", ["-G", "6000000", "-H", "5", "-c", "3", "-n", "2000", "-s", "35"])
run("no cigar", ["-G", "3000000", "-H", "3", "-n", "2000", "-s", "36"], cigar=False)
run("big 20k reads", ["-G", "20000000", "-H", "3", "-n", "20000", "-s", "37"])
PY
