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
    print(json.dumps({"source": source, "kernels": dict(sorted(ker.items()))}, indent=1))


def main():
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
