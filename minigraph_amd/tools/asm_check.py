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
    ap.add_argument("--no-ref", action="store_true", help="skip the reference run (timing iterations)")
    ap.add_argument("--err", default="0.001")
    ap.add_argument("--keep-ref", default=None, help="a directory in which the workload and the reference's GAF are kept and reused (several runs against one reference run)")
    a = ap.parse_args()
    mga.load()
    if a.keep_ref:
        d = os.path.join(a.keep_ref, "G%d_c%d_H%d_n%d_l%d_e%s" % (a.genome, a.chr, a.hap, a.n, a.contig, a.err))
        os.makedirs(d, exist_ok=True)
    else:
        d = tempfile.mkdtemp(prefix="mga_asm_")
    try:
        if not os.path.exists(os.path.join(d, "a.reads.fa.done")):
            subprocess.check_call([mga.MGSIM, "-p", os.path.join(d, "a"), "-G", str(a.genome), "-c", str(a.chr), "-H", str(a.hap), "-n", str(a.n), "-l", str(a.contig), "-e", a.err, "-s", "5"],
                                  stderr=subprocess.DEVNULL)
            open(os.path.join(d, "a.reads.fa.done"), "w").close()
        g, r = os.path.join(d, "a.gfa"), os.path.join(d, "a.reads.fa")
        ok = True
        for cigar in ((True,) if a.cigar_only else (False, True)):
            got, ref = os.path.join(d, "got.gaf"), os.path.join(d, "ref%d.gaf" % int(cigar))
            t0 = time.time()
            mga.map_files(g, [r], got, preset="asm", cigar=cigar, n_threads=a.threads)
            t1 = time.time()
            out = {"graph_backbone_bp": a.genome, "chr": a.chr, "hap": a.hap, "contig_bp": a.contig, "n": a.n, "query_bp": os.path.getsize(r), "cigar": cigar, "threads": a.threads,
                   "t_ours_total_s": round(t1 - t0, 2), "gaf_bytes": os.path.getsize(got), "note": "file -> file, graph load + index build included on both sides"}
            import ctypes
            st = (ctypes.c_int64 * 8)()
            mga.load().mga_rq_dev_stats(st, 1)
            out["rmq_runs"] = dict(device=st[0], host_tie=st[1], host_inner_window=st[2], host_long=st[3], host_device_failed=st[4])
            if os.path.exists(REF_BIN) and not (a.no_ref and not os.path.exists(ref + ".done")):
                if not os.path.exists(ref + ".done"):
                    with open(ref, "wb") as fo:
                        subprocess.check_call([REF_BIN] + (["-c"] if cigar else []) + ["-x", "asm", "-t", str(a.threads), g, r], stdout=fo, stderr=subprocess.DEVNULL)
                    out["t_ref_total_s"] = round(time.time() - t1, 2)
                    if a.keep_ref:
                        open(ref + ".done", "w").write(str(out["t_ref_total_s"]))
                else:
                    out["t_ref_total_s"] = float(open(ref + ".done").read())
                same = subprocess.call(["cmp", "-s", got, ref]) == 0
                out["parity"] = "GAF byte-identical to the reference" if same else "MISMATCH"
                ok = ok and same
            print(json.dumps(out), flush=True)
        sys.exit(0 if ok else 1)
    finally:
        if not a.keep_ref:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
