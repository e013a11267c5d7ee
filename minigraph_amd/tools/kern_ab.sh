#!/bin/bash
# isolated per-kernel times of the bench workload under a set of environment knobs:  kern_ab.sh <tag> [ENV=VALUE ...]
tag=$1; shift
B="python bench.py --steps 2 --warmup 1 --no-cpu --resident-steps 1 --one-placement --no-asm --no-small --no-file-out --no-rank-share"
env "$@" $B > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err
python - <<P
import json; d=json.load(open("gpurun_out/${tag}.json")); k=d["kernels_ms_isolated"]
print("${tag}", "$*", "value %.3f Gbp/s" % d["value"], {n: k[n] for n in ("k_sketch","k_seed_fill","k_lchain","k_text","k_wfa_tb","k_gaf") if n in k}, "sum %.1f ms" % sum(k.values()))
P
