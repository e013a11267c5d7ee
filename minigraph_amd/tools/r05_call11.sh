#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_stages.py -q -x -m gpu -k "wfa or lchain" 2>&1 | tail -4
MGA_LC_AOS=1 timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_e2e.py -q -x -m gpu -k "lchain or mt_known or synthetic_vs_reference or parity_sweep or long_join or command_line" 2>&1 | tail -4 | tee $out/r05k_tests_aos.txt
echo "[tests] $(( $(date +%s) - t0 )) s"
WD=/tmp/mga_wd
python bench.py --steps 3 --warmup 1 --workdir $WD --no-cpu --resident-steps 0 --one-placement --no-asm --no-rank-share > /dev/null 2>&1
STEPS=8 RESIDENT=1 BENCH_ARGS="--workdir $WD --no-asm --no-rank-share" bash minigraph_amd/tools/knob_sweep.sh - "MGA_LC_AOS=1" - "MGA_LC_AOS=1" 2>&1 | tee $out/r05k_aos_sweep.txt
echo "[sweep] $(( $(date +%s) - t0 )) s"
