#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
MGA_WFA_PACKED=1 timeout 600 python -m pytest tests/test_gpu_stages.py -q -x -m gpu -k "wfa" 2>&1 | tail -15 | tee $out/r05l_tests_wfa_packed.txt
echo "[tests] $(( $(date +%s) - t0 )) s"
MGA_WFA_PACKED=1 timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "mt_known or synthetic_vs_reference or parity_sweep" 2>&1 | tail -8 | tee $out/r05l_tests_e2e_packed.txt
echo "[tests e2e] $(( $(date +%s) - t0 )) s"
WD=/tmp/mga_wd
python bench.py --steps 3 --warmup 1 --workdir $WD --no-cpu --resident-steps 0 --one-placement --no-asm --no-rank-share > /dev/null 2>&1
STEPS=6 RESIDENT=1 BENCH_ARGS="--workdir $WD --no-asm --no-rank-share" bash minigraph_amd/tools/knob_sweep.sh - "MGA_WFA_PACKED=1" - "MGA_WFA_PACKED=1" 2>&1 | tee $out/r05l_packed_sweep.txt
echo "[sweep] $(( $(date +%s) - t0 )) s"
