#!/usr/bin/env python3
"""-x asm check on chromosome-scale queries: contigs sampled from the haplotype walks of a 50 Mbp bubble graph
(0.1 % divergence), mapped file -> file with and without -c through mg_map_files() on the GPU and through the unmodified reference
binary (oracle/_ref/minigraph, test infrastructure); the GAF files must be the same bytes.  Prints one JSON line per run.

    python minigraph_amd/tools/asm_check.py --contig 50000000 --n 2"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import minigraph_amd as mga  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "minigraph")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contig", type=int, default=50000000)
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--genome", type=int, default=50000000, help="backbone bp of the graph")
    ap.add_argument("--chr", type=int, default=1)
    ap.add_argument("--hap", type=int, default=3)
    ap.add_argument("--cigar-only", action="store_true", help="only the -c run (BASELINE configs[4] is -cx asm)")
    a = ap.parse_args()
    mga.load()
    d = tempfile.mkdtemp(prefix="mga_asm_")
    try:
        subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "a"), "-G", str(a.genome), "-c", str(a.chr), "-H", str(a.hap), "-n", str(a.n), "-l", str(a.contig), "-e", "0.001", "-s", "5"],
                              stderr=subprocess.DEVNULL)
        g, r = os.path.join(d, "a.gfa"), os.path.join(d, "a.reads.fa")
        ok = True
        for cigar in ((True,) if a.cigar_only else (False, True)):
            got, ref = os.path.join(d, "got.gaf"), os.path.join(d, "ref.gaf")
            t0 = time.time()
            mga.map_files(g, [r], got, preset="asm", cigar=cigar, n_threads=a.threads)
            t1 = time.time()
            out = {"graph_backbone_bp": a.genome, "chr": a.chr, "hap": a.hap, "contig_bp": a.contig, "n": a.n, "query_bp": os.path.getsize(r), "cigar": cigar, "threads": a.threads,
                   "t_ours_total_s": round(t1 - t0, 2), "gaf_bytes": os.path.getsize(got), "note": "file -> file, graph load + index build included on both sides"}
            if os.path.exists(REF_BIN):
                with open(ref, "wb") as fo:
                    subprocess.check_call([REF_BIN] + (["-c"] if cigar else []) + ["-x", "asm", "-t", str(a.threads), g, r], stdout=fo, stderr=subprocess.DEVNULL)
                out["t_ref_total_s"] = round(time.time() - t1, 2)
                same = subprocess.call(["cmp", "-s", got, ref]) == 0
                out["parity"] = "GAF byte-identical to the reference" if same else "MISMATCH"
                ok = ok and same
            print(json.dumps(out), flush=True)
        sys.exit(0 if ok else 1)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
