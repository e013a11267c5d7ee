#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
WD=/tmp/mga_wd
python bench.py --steps 3 --warmup 1 --workdir $WD --no-cpu --resident-steps 0 --one-placement --no-asm --no-rank-share > /dev/null 2>&1
export STEPS=10 BENCH_ARGS="--workdir $WD --no-asm --no-rank-share"
bash minigraph_amd/tools/knob_sweep.sh - "MGA_JOBRAMP=0" "MGA_JOBRAMP=2" "MGA_JOBRAMP=4" - "MGA_JOBRAMP=0" "MGA_JOBRAMP=3 MGA_TAIL=2" "MGA_JOBRAMP=3 MGA_FRONT_SLOTS=2" "MGA_JOBRAMP=3 MGA_FRONT_SLOTS=2 MGA_PIPE=6" - 2>&1 | tee $out/r05h_ramp_sweep.txt
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "knobs or mt_known or several_query" 2>&1 | tail -3
