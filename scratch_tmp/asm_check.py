import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import minigraph_amd as mga, refbind as rb
mga.load()
d = tempfile.mkdtemp()
L = int(sys.argv[1]); n = int(sys.argv[2])
subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "a"), "-G", "50000000", "-H", "3", "-n", str(n), "-l", str(L), "-e", "0.001", "-s", "5"], stderr=subprocess.DEVNULL)
g, r = os.path.join(d, "a.gfa"), os.path.join(d, "a.reads.fa")
for cigar in (False, True):
    t0 = time.time(); mga.map_files(g, [r], os.path.join(d, "got.gaf"), preset="asm", cigar=cigar, n_threads=16); t1 = time.time()
    with open(os.path.join(d, "ref.gaf"), "wb") as fo:
        subprocess.check_call([rb.REF_BIN] + (["-c"] if cigar else []) + ["-x", "asm", "-t", "16", g, r], stdout=fo, stderr=subprocess.DEVNULL)
    t2 = time.time()
    same = open(os.path.join(d, "got.gaf"), "rb").read() == open(os.path.join(d, "ref.gaf"), "rb").read()
    print("contig", L, "n", n, "cigar", cigar, "ours %.2fs ref %.2fs" % (t1 - t0, t2 - t1), "SAME" if same else "DIFF", flush=True)
