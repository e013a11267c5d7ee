import os, sys, time, subprocess, tempfile
sys.path.insert(0, os.getcwd())
import minigraph_amd as mga
d = tempfile.mkdtemp()
subprocess.run([mga.MGSIM, "-p", os.path.join(d, "w"), "-G", "2350000000", "-c", "24", "-H", "5", "-n", "10", "-s", "11"], check=True, stderr=subprocess.DEVNULL)
os.remove(os.path.join(d, "w.lin.fa"))
G = mga.Graph(os.path.join(d, "w.gfa"), n_threads=16)
G.save_image(os.path.join(d, "w.mgi")); G.close()
for it in range(3):
    t0 = time.time(); G = mga.Graph(os.path.join(d, "w.mgi"), n_threads=16, image=True); print("load %.3f s" % (time.time() - t0), flush=True); G.close()
