#!/usr/bin/env python3
"""Chunk-size fault hunt: one synthetic workload, mapped file -> GAF in memory under several pipeline settings, each in its
own process (a crash must not take the others down); the md5 of every run is compared with the default-setting run.

    python minigraph_amd/tools/chunk_repro.py --genome 600000000 --reads 50000 --out gpurun_out/chunk_repro.jsonl CASE [CASE ...]

CASE = comma-separated NAME=VALUE environment settings, e.g.  MGA_CHUNK=32768,MGA_DEV_GCHAIN=0,MGA_PIPE=1   ("default" = none)."""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(graph, reads, threads):
    import minigraph_amd as mga
    G = mga.Graph(graph, preset="lr", cigar=True, n_threads=threads)
    t0 = time.time()
    m = mga.map_files_idx(G, [reads], n_threads=threads)
    dt = time.time() - t0
    print(json.dumps(dict(md5=hashlib.md5(m.view().tobytes()).hexdigest(), bytes=len(m), map_s=round(dt, 3), stats={k: int(v) for k, v in mga.get_stats(G).items() if isinstance(v, int)})), flush=True)
    m.free()
    G.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=600000000)
    ap.add_argument("--chr", type=int, default=8)
    ap.add_argument("--hap", type=int, default=5)
    ap.add_argument("--reads", type=int, default=50000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--timeout", type=int, default=240)
    ap.add_argument("--out", default="")
    ap.add_argument("--child", nargs=2)
    ap.add_argument("cases", nargs="*")
    a = ap.parse_args()
    if a.child:
        child(a.child[0], a.child[1], a.threads)
        return
    import minigraph_amd as mga
    mga.load()
    d = tempfile.mkdtemp(prefix="mga_chunk_")
    subprocess.run([mga.MGSIM, "-p", os.path.join(d, "g"), "-G", str(a.genome), "-c", str(a.chr), "-H", str(a.hap), "-n", str(a.reads), "-s", "11"],
                   stderr=subprocess.PIPE, check=True)
    os.remove(os.path.join(d, "g.lin.fa"))
    graph, reads = os.path.join(d, "g.gfa"), os.path.join(d, "g.reads.fa")
    base = None
    fo = open(a.out, "a") if a.out else None
    for case in ["default"] + list(a.cases):
        env = dict(os.environ)
        if case != "default":
            for kv in case.split(","):
                k, v = kv.split("=", 1)
                env[k] = v
        env.setdefault("MGA_SEGV_TRACE", "1")
        t0 = time.time()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--threads", str(a.threads), "--child", graph, reads], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=a.timeout)
            rc, so, se = p.returncode, p.stdout.decode(), p.stderr.decode()
        except subprocess.TimeoutExpired as e:
            rc, so, se = -999, (e.stdout or b"").decode(), (e.stderr or b"").decode()
        rec = dict(case=case, rc=rc, wall_s=round(time.time() - t0, 1))
        m = re.search(r"^\{.*\}$", so, re.M)
        if m:
            rec.update(json.loads(m.group(0)))
            if case == "default":
                base = rec.get("md5")
            rec["same_as_default"] = rec.get("md5") == base
        if rc != 0 or not m:
            rec["stderr_tail"] = se[-600:]
            if a.out:
                open(a.out + "." + re.sub(r"[^A-Za-z0-9=]+", "_", case)[:80] + ".err", "w").write(se[:200000])
        line = json.dumps(rec)
        print(line, flush=True)
        if fo:
            fo.write(line + "\n")
            fo.flush()


if __name__ == "__main__":
    main()
