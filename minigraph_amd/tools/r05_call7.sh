#!/bin/bash
# round 5, GPU call 7: 200 kb WFA pair, per-stream tokens, then BASELINE configs[4] at its size on one GPU: one 3 Gbp graph, 24 contigs of 98 Mbp (2.35 Gbp of query), -cx asm
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_stages.py -q -x -m gpu -k "chained_fallback" 2>&1 | tail -5 | tee $out/r05g_tests_wfa.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "knobs or mt_known or chunks_beyond or deterministic" 2>&1 | tail -5 | tee $out/r05g_tests_e2e.txt
echo "[tests] $(( $(date +%s) - t0 )) s"
WD=/tmp/mga_wd
python bench.py --steps 4 --warmup 1 --workdir $WD --no-cpu --resident-steps 0 --no-asm > $out/r05g_bench.json 2> $out/r05g_bench.err; tail -2 $out/r05g_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05g_bench.json"))
print("value %.3f  device_placement %s  rank_share %s" % (d["value"], {k: d.get("device_placement",{}).get(k) for k in ("value","cpu_s_per_step")}, {k: d.get("rank_share",{}).get(k) for k in ("value","cpu_s_per_step","vs_device_placement")}))
PY
MGA_PIPE=4 python bench.py --steps 4 --warmup 1 --workdir $WD --no-cpu --resident-steps 0 --no-asm > $out/r05g_bench_pipe4.json 2> /dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05g_bench_pipe4.json"))
print("MGA_PIPE=4: value %.3f  device_placement %s  rank_share %s" % (d["value"], {k: d.get("device_placement",{}).get(k) for k in ("value","cpu_s_per_step")}, {k: d.get("rank_share",{}).get(k) for k in ("value","cpu_s_per_step","vs_device_placement")}))
PY
echo "[bench] $(( $(date +%s) - t0 )) s"
MGA_DEBUG_PIPE=1 timeout 900 python minigraph_amd/tools/asm_check.py --genome 2350000000 --chr 24 --hap 5 --n 24 --contig 98000000 --cigar-only > $out/r05g_asm_config5.txt 2> $out/r05g_asm_config5.err; tail -1 $out/r05g_asm_config5.txt; grep "\[rq\] [0-9]" $out/r05g_asm_config5.err | tail -8
echo "[config 5] $(( $(date +%s) - t0 )) s"
