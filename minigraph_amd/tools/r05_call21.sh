#!/bin/bash
# index-ordered work lists for the narrow WFA rungs (MGA_WFA_LIST_STABLE, default 3): WFA stage tests + e2e subset, then A/B sweep
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 200 python -u -m pytest tests/test_gpu_stages.py -q -x -m gpu -k "wfa" 2>&1 | tail -4 | tee $out/r05r_tests_wfa.txt
rc=${PIPESTATUS[0]}; echo "[tests] rc $rc $(( $(date +%s) - t0 )) s"; [ $rc -ne 0 ] && exit 0
timeout 150 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "mt_known or synthetic_vs_reference or parity_sweep or knobs" 2>&1 | tail -4 | tee $out/r05r_tests_e2e.txt
rc=${PIPESTATUS[0]}; echo "[tests e2e] rc $rc $(( $(date +%s) - t0 )) s"; [ $rc -ne 0 ] && exit 0
WD=/tmp/mga_wd
timeout 60 python bench.py --steps 3 --warmup 1 --workdir $WD --no-cpu --resident-steps 0 --one-placement --no-asm --no-rank-share > /dev/null 2>&1
STEPS=6 RESIDENT=1 BENCH_ARGS="--workdir $WD --no-asm --no-rank-share" timeout 200 bash minigraph_amd/tools/knob_sweep.sh - "MGA_WFA_LIST_STABLE=0" "MGA_WFA_LIST_STABLE=2" - 2>&1 | tee $out/r05r_list_sweep.txt
echo "[sweep] $(( $(date +%s) - t0 )) s"
