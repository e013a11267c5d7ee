#!/bin/bash
# rocprofv3 kernel trace of one bench command; summaries land in gpurun_out/<tag>_*.  usage: prof_trace.sh <tag> [env assignments...] -- <bench args>
# (run on the GPU box through gpurun; kernel-trace and --pmc passes are separate runs, never combined)
set -u
tag=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
env "${envs[@]}" rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python bench.py "$@" > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_rocprof.err
python - "$out" "$tag" <<'PY'
import csv, glob, sys, collections, re
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
if not f:
    print("no kernel trace found"); sys.exit(0)
d = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
with open("gpurun_out/%s_kernel_stats.txt" % tag, "w") as fo:
    fo.write("# rocprofv3 --kernel-trace: per kernel launches, total ms, average / median / max us, share\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        fo.write("%-70s %6d %10.2f %10.1f %10.1f %10.1f %6.2f%%\n" % (k[:70], len(v), sum(v) / 1e3, sum(v) / len(v), v2[len(v2) // 2], v2[-1], 100 * sum(v) / tot))
print(open("gpurun_out/%s_kernel_stats.txt" % tag).read())
PY
