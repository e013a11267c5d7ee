"""Debugging aid (GPU box): one small workload through graph chaining on the device, one-kernel form vs three-kernel form, each in a child under a short timeout.
  python minigraph_amd/tools/gc_split_dbg.py [mgsim arguments]"""
import hashlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CHILD = r"""
import os, sys, hashlib
sys.path.insert(0, %r)
import minigraph_amd as mga
G = mga.Graph(sys.argv[1], n_threads=4)
m = mga.map_files_idx(G, [sys.argv[2]], n_threads=4)
b = bytes(m.bytes()); m.free()
print("OUT", len(b), hashlib.md5(b).hexdigest(), mga.get_stats(G)["n_gc_retry"])
""" % ROOT
import minigraph_amd as mga
simargs = sys.argv[1:] or ["-G", "3000000", "-H", "5", "-n", "400", "-l", "12000", "-s", "43"]   # arguments of mgsim
d = tempfile.mkdtemp()
subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "t")] + simargs, stderr=subprocess.DEVNULL)
for tag, env in (("mono", {"MGA_GC_SPLIT": "0"}), ("split", {"MGA_GC_SPLIT": "1", "MGA_GC_SPLIT_DEBUG": os.environ.get("DBG_LEVEL", "1")})):
    e = dict(os.environ, MGA_DEV_GCHAIN="1", MGA_PIPE="1", **env)
    try:
        p = subprocess.run([sys.executable, "-c", CHILD, os.path.join(d, "t.gfa"), os.path.join(d, "t.reads.fa")], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=int(os.environ.get("DBG_TIMEOUT", "60")))
        print(tag, "rc", p.returncode, [l for l in p.stdout.decode().splitlines() if l.startswith("OUT")], flush=True)
        print("\n".join(l for l in p.stderr.decode().splitlines() if "gc-split" in l or "rror" in l or "fault" in l.lower())[-3000:], flush=True)
    except subprocess.TimeoutExpired as ex:
        print(tag, "TIMEOUT", flush=True)
        print((ex.stderr or b"").decode()[-3000:], flush=True)
