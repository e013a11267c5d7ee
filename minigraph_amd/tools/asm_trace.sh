#!/bin/bash
# rocprofv3 kernel trace of one -cx asm job (10 x 50 Mbp contigs vs a 500 Mbp graph): which kernels the 2 s "wfa" phase of a batch consists of.  usage (GPU box): asm_trace.sh <tag>
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
d=$(mktemp -d)
minigraph_amd/lib/mgsim -p $d/a -G 500000000 -c 10 -H 3 -n 10 -l 50000000 -e 0.001 -s 5 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python -c "
import sys; sys.path.insert(0, '.')
import minigraph_amd as mga
mga.map_files('$d/a.gfa', ['$d/a.reads.fa'], '$d/got.gaf', preset='asm', cigar=True, n_threads=16)
" > gpurun_out/${tag}_run.txt 2> gpurun_out/${tag}_rocprof.err
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
    fo.write("# rocprofv3 --kernel-trace of one -cx asm job, 10 x 50 Mbp contigs vs a 500 Mbp graph: per kernel launches, total ms, average / median / max us, share\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        fo.write("%-70s %6d %10.2f %10.1f %10.1f %10.1f %6.2f%%\n" % (k[:70], len(v), sum(v) / 1e3, sum(v) / len(v), v2[len(v2) // 2], v2[-1], 100 * sum(v) / tot))
print(open("gpurun_out/%s_kernel_stats.txt" % tag).read()[:3500])
PY
rm -rf "$out" "$d"
