#!/bin/bash
# host CPU seconds by stage of two steps of the bench workload (MGA_DEBUG_PIPE=1):  cpu_ab.sh <tag> [ENV=VALUE ...]
tag=$1; shift
env MGA_DEBUG_PIPE=1 "$@" python bench.py --steps 2 --warmup 1 --no-cpu --resident-steps 0 --one-placement --no-asm --no-small --no-file-out --no-rank-share > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err
python - <<P
import json,re
d=json.load(open("gpurun_out/${tag}.json"))
L=[l for l in open("gpurun_out/${tag}.err") if l.startswith("[pipe] host CPU seconds")]
print("${tag}", "$*", "value %.3f" % d["value"], "cpu_s_per_step", d["host"]["cpu_s_per_step"])
print("   ", re.sub(r".*done at [0-9.]+ s\):", "", L[-1]).strip()[:400])
P
