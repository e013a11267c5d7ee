#!/bin/bash
# round 5, GPU call 6: multi-read device RMQ launches + GAF lines written straight to their place in the output
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "rmq_forward or asm_preset_long or both_presets or mt_known or knobs or parity_sweep or edge_case or several_query or asm_200Mbp" 2>&1 | tail -15 | tee $out/r05f_tests.txt
echo "[tests] $(( $(date +%s) - t0 )) s"
A="--genome 500000000 --chr 10 --n 10 --contig 50000000 --cigar-only --keep-ref /tmp/asm_wd"
MGA_DEBUG_PIPE=1 python minigraph_amd/tools/asm_check.py $A > $out/r05f_asm_dev.txt 2> $out/r05f_asm_dev.err; tail -1 $out/r05f_asm_dev.txt; grep "\[rq\]" $out/r05f_asm_dev.err | tail -12
python minigraph_amd/tools/asm_check.py $A --no-ref > $out/r05f_asm_dev2.txt 2> /dev/null; tail -1 $out/r05f_asm_dev2.txt
echo "[asm] $(( $(date +%s) - t0 )) s"
WD=/tmp/mga_wd
python bench.py --steps 3 --warmup 1 --workdir $WD --no-cpu --resident-steps 0 --one-placement --no-asm --no-rank-share > $out/r05f_first.json 2> $out/r05f_first.err; tail -3 $out/r05f_first.err
export STEPS=8 BENCH_ARGS="--workdir $WD --no-asm --no-rank-share"
bash minigraph_amd/tools/knob_sweep.sh - "MGA_GAF_DIRECT=0" - "MGA_GAF_DIRECT=0" - 2>&1 | tee $out/r05f_gaf_sweep.txt
BENCH_ARGS="$BENCH_ARGS --placement device --threads 2" bash minigraph_amd/tools/knob_sweep.sh - "MGA_GAF_DIRECT=0" "MGA_PIPE=4" "MGA_PIPE=4 MGA_FRONT_SLOTS=2" - 2>&1 | tee $out/r05f_share_sweep.txt
echo "[sweeps] $(( $(date +%s) - t0 )) s"
