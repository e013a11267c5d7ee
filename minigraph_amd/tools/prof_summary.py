#!/usr/bin/env python
"""Turn a rocprofv3 `--kernel-trace --stats` result (sqlite .db or *_kernel_stats.csv) into the short
text summary committed under profiles/.\nusage: prof_summary.py <results.db|kernel_stats.csv> [title]   |   prof_summary.py --pmc <results.db> [title]   |   prof_summary.py --pmc-json <fetch.db> <write.db> [source]"""
import csv
import sqlite3
import sys


def rows(path):
    if path.endswith(".db"):
        cur = sqlite3.connect(path).cursor()
        for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            yield name, int(calls), float(tot), float(avg), float(pct)
    else:
        for r in csv.DictReader(open(path)):
            yield r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])


def pmc_rows(path):
    """per kernel: launches, total and per-launch KB of one --pmc pass (FETCH_SIZE / WRITE_SIZE are reported in KB)"""
    cur = sqlite3.connect(path).cursor()
    q = "select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name order by sum(value) desc"
    for name, counter, calls, tot in cur.execute(q):
        yield name, counter, int(calls), float(tot)


def main_pmc(path, title):
    print("# rocprofv3 --pmc (separate pass, no tracing)  %s" % title)
    print("# %-40s %-12s %8s %14s %14s" % ("kernel", "counter", "launches", "total_KB", "KB_per_launch"))
    for name, counter, calls, tot in pmc_rows(path):
        print("%-42s %-12s %8d %14.1f %14.1f" % (name.split("(")[0][:42], counter, calls, tot, tot / max(1, calls)))


def main_pmc_json(fetch_db, write_db, source):
    """profiles/*_pmc.json as bench.py reads it: per kernel the KB and launch counts of the FETCH_SIZE and the WRITE_SIZE pass"""
    import json
    ker = {}
    for path, key in ((fetch_db, "fetch"), (write_db, "write")):
        for name, counter, calls, tot in pmc_rows(path):
            k = ker.setdefault(name.split("(")[0].replace("void ", "").strip(), {})
            k[key + "_kb"] = round(k.get(key + "_kb", 0.0) + tot, 1)
            k["launches_" + key] = k.get("launches_" + key, 0) + calls
    import glob, hashlib, os
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    h = hashlib.sha1()   # which WFA sources the passes ran (bench.py labels the traffic figure STALE when the tree has others)
    for f in sorted(glob.glob(os.path.join(root, "minigraph_amd", "csrc", "k_wfa*.hip")) + [os.path.join(root, "minigraph_amd", "csrc", "wfa_window.h")]):
        h.update(open(f, "rb").read())
    print(json.dumps({"source": source, "wfa_src_sha1": h.hexdigest(), "kernel_src_sha1": kernel_src_sha1(), "kernels": dict(sorted(ker.items()))}, indent=1))


def kernel_src_sha1():
    """hash of every device source of the tree (csrc/*.hip and the headers they include): what a counter / PMC file records about the kernels it was taken from, and what
    bench.py compares with the tree it runs in (a file of other kernels is labelled STALE, whichever family it prices)"""
    import glob, hashlib, os
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(root, "minigraph_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "minigraph_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()


def main_sq(paths, title):
    """per kernel, over one or more --pmc passes of SQ counters: instructions per wave and the shares of the waves' resident cycles.  Every pass is normalised by ITS OWN
    SQ_WAVES / SQ_WAVE_CYCLES / SQ_BUSY_CYCLES (all passes collect the first two), then the columns of the passes are put side by side."""
    acc = {}
    for path in paths:
        one = {}
        for name, counter, calls, tot in pmc_rows(path):
            k = one.setdefault(name.split("(")[0].replace("void ", "").strip()[:46], {})
            k[counter] = k.get(counter, 0.0) + tot
            k["_calls"] = calls
        for name, k in one.items():
            w, cyc, busy = k.get("SQ_WAVES", 0.0), k.get("SQ_WAVE_CYCLES", 0.0), k.get("SQ_BUSY_CYCLES", 0.0) * 32.0
            d = acc.setdefault(name, {})
            d.setdefault("calls", k.get("_calls", 0)); d.setdefault("waves", w); d.setdefault("cycles", cyc)
            for c, v in k.items():
                if c.startswith("_") or c in ("SQ_WAVES", "SQ_WAVE_CYCLES"):
                    continue
                if c.startswith("SQ_INSTS_") and w:
                    d.setdefault(c + "/wave", v / w)
                elif cyc:
                    d.setdefault(c + "/cyc", v / cyc)
            if busy and cyc:
                d.setdefault("occ", cyc / busy)
                if "SQ_ACTIVE_INST_VALU" in k:
                    d.setdefault("valu_busy", k["SQ_ACTIVE_INST_VALU"] / busy)
    print("# rocprofv3 --pmc SQ_* (own passes, no tracing)  %s" % title)
    print("# kernel_src_sha1: %s" % kernel_src_sha1())
    import re
    m = re.search(r"--reads (\d+)", title)
    reads = int(m.group(1)) if m else 125000
    lc = acc.get("k_lchain<6>") or acc.get("k_lchain<7>") or acc.get("k_lchain<5>") or acc.get("k_lchain")
    if lc and lc.get("waves"):   # k_lchain is one wavefront per read: its waves / the reads of a pass = the passes these counters cover (bench.py divides by it)
        print("# passes covered: %d (k_lchain: %d waves / %d reads per pass)" % (round(lc["waves"] / reads), lc["waves"], reads))
    print("# per wave: instructions issued; shares: fraction of the waves' resident cycles (SQ_WAVE_CYCLES): valu = SQ_ACTIVE_INST_VALU, wait = SQ_WAIT_ANY (parked on s_waitcnt / barrier),")
    print("# stall = SQ_WAIT_INST_ANY (issue stalls), vmem / lds / salu = SQ_INST_CYCLES_VMEM / SQ_ACTIVE_INST_LDS / SQ_INST_CYCLES_SALU where collected")
    print("# occ = waves resident per SIMD while the kernel runs = SQ_WAVE_CYCLES / (32 x SQ_BUSY_CYCLES) (SQ_BUSY_CYCLES is summed over the 32 shader engines, each with 32 SIMDs);")
    print("# valu_busy = SQ_ACTIVE_INST_VALU / (32 x SQ_BUSY_CYCLES).  CALIBRATION (profiles/r05a_valu_calibration.txt, a kernel of known instruction count under these counters):")
    print("#   SQ_INSTS_VALU is exact; SQ_ACTIVE_INST_VALU equals it (it counts issue slots of FOUR cycles, so the `valu` and `valu_busy` columns are a quarter of a cycle share, over")
    print("#   durations inflated by the counter collection); SQ_WAVE_CYCLES is in the same units and saturates with occupancy.  The utilisation figure that holds is bench.py's")
    print("#   roofline.valu_busy: VALU/wave x waves / passes here, over 578 M wave-instructions per second per SIMD [measured] and the kernel's time WITHOUT counters")
    print("%-46s %7s %12s %11s %11s %9s %7s %7s %7s %7s %7s %7s %6s %9s" % ("kernel", "calls", "waves", "VALU/wave", "SALU/wave", "LDS/wave", "valu", "wait", "stall", "vmem", "lds", "salu", "occ", "valu_busy"))

    def col(d, key, fmt, width):
        return (fmt % d[key]) if key in d else " " * (width - 1) + "-"
    for name, d in sorted(acc.items(), key=lambda kv: -kv[1].get("cycles", 0)):
        if not d.get("waves"):
            continue
        print("%-46s %7d %12d %s %s %s %s %s %s %s %s %s %s %s" % (name, d["calls"], d["waves"], col(d, "SQ_INSTS_VALU/wave", "%11.0f", 11), col(d, "SQ_INSTS_SALU/wave", "%11.0f", 11),
              col(d, "SQ_INSTS_LDS/wave", "%9.0f", 9), col(d, "SQ_ACTIVE_INST_VALU/cyc", "%7.3f", 7), col(d, "SQ_WAIT_ANY/cyc", "%7.3f", 7), col(d, "SQ_WAIT_INST_ANY/cyc", "%7.3f", 7),
              col(d, "SQ_INST_CYCLES_VMEM/cyc", "%7.3f", 7), col(d, "SQ_ACTIVE_INST_LDS/cyc", "%7.3f", 7), col(d, "SQ_INST_CYCLES_SALU/cyc", "%7.3f", 7), col(d, "occ", "%6.2f", 6), col(d, "valu_busy", "%9.3f", 9)))


def main_calib(path):
    """valu_rate under --pmc (csv): per dispatch of the microbenchmark -- whose instruction count (64 x iters + loop overhead per wave), waves per SIMD and issue rate
    (an independent stream: ~100 %) are KNOWN -- the raw counters and what prof_summary's formulas make of them: pins the counters' units"""
    import collections
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]), int(r.get("Workgroup_Size", 64) or 64))
        e = d.setdefault(k, {})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            e["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("# valu_rate under rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE: one row per dispatch")
    print("# expected: VALU/wave = 64 x iters (+ a few), waves/SIMD = grid / 64 / 1024; an independent instruction stream keeps the vector ALU busy ~100 %")
    print("%-16s %8s %6s %10s %12s %14s %14s %14s %14s %10s | %9s %9s %9s %9s" % ("kernel", "waves", "w/SIMD", "dur_us", "VALU/wave", "ACTIVE_VALU", "WAVE_CYCLES", "BUSY_CYCLES", "GRBM_ACTIVE", "GRBM_MHz",
                                                                                     "occ(x32)", "busy(x32)", "WC/wave/us", "ACT/INST"))
    for (did, name, grid, wg), e in d.items():
        waves = e.get("SQ_WAVES", 0.0) or (grid / 64.0)
        us = e.get("_ns", 0) / 1e3
        wc, bc, av, iv, ga = e.get("SQ_WAVE_CYCLES", 0.0), e.get("SQ_BUSY_CYCLES", 0.0), e.get("SQ_ACTIVE_INST_VALU", 0.0), e.get("SQ_INSTS_VALU", 0.0), e.get("GRBM_GUI_ACTIVE", 0.0)
        print("%-16s %8d %6.2f %10.1f %12.1f %14.0f %14.0f %14.0f %14.0f %10.1f | %9.3f %9.3f %9.1f %9.3f" % (
            name[:16], waves, grid / 64.0 / 1024.0, us, iv / max(waves, 1), av, wc, bc, ga, ga / us if us else 0.0,
            wc / (32.0 * bc) if bc else 0.0, av / (32.0 * bc) if bc else 0.0, wc / max(waves, 1) / us if us else 0.0, av / iv if iv else 0.0))


def main():
    if sys.argv[1] == "--calib":
        return main_calib(sys.argv[2])
    if sys.argv[1] == "--sq":
        return main_sq([a for a in sys.argv[2:] if a.endswith(".db")], " ".join(a for a in sys.argv[2:] if not a.endswith(".db")))
    if sys.argv[1] == "--pmc-json":
        return main_pmc_json(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
    if sys.argv[1] == "--pmc":
        return main_pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
    path, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    print("# rocprofv3 --kernel-trace --stats  %s" % title)
    print("# %-40s %8s %14s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows(path):
        print("%-42s %8d %14.1f %14.1f %7.2f" % (name.split("(")[0][:42], calls, tot, avg, pct))


if __name__ == "__main__":
    main()
