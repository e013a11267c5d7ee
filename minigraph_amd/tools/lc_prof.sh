#!/bin/bash
# per-phase cycle counts of k_lchain on the bench workload (isolated pass):  lc_prof.sh <tag> [ENV=VALUE ...]
tag=$1; shift
B="python bench.py --steps 1 --warmup 1 --no-cpu --resident-steps 1 --one-placement --no-asm --no-small --no-file-out --no-rank-share"
env MGA_LC_PROF=1 MGA_PIPE=1 MGA_WFA_SIDE=0 "$@" $B > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err
grep "lc-prof" gpurun_out/${tag}.err | tail -1
python - <<P
import json; d=json.load(open("gpurun_out/${tag}.json")); print("${tag}", "k_lchain", d["kernels_ms_isolated"].get("k_lchain"), "value", d["value"])
P
