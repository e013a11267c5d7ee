#!/bin/bash
# round 5, GPU call 10: tiled traceback layout + reachable-only rows: parity, isolated kernel times, FETCH / WRITE passes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_stages.py -q -x -m gpu -k "wfa" 2>&1 | tail -5 | tee $out/r05j_tests_wfa.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "mt_known or synthetic_vs_reference or parity_sweep or gap_beyond" 2>&1 | tail -5 | tee $out/r05j_tests_e2e.txt
echo "[tests] $(( $(date +%s) - t0 )) s"
WD=/tmp/mga_wd
python bench.py --steps 3 --warmup 1 --workdir $WD --no-cpu --resident-steps 0 --one-placement --no-asm --no-rank-share > /dev/null 2>&1
STEPS=8 RESIDENT=1 BENCH_ARGS="--workdir $WD --no-asm --no-rank-share" bash minigraph_amd/tools/knob_sweep.sh - - 2>&1 | tee $out/r05j_sweep.txt
echo "[sweep] $(( $(date +%s) - t0 )) s"
PROF_PARTS="pmc" bash minigraph_amd/tools/prof_all.sh r05j > $out/r05j_prof_all.log 2>&1
head -30 $out/r05j_pmc_write.txt; head -12 $out/r05j_pmc_fetch.txt
echo "[pmc] $(( $(date +%s) - t0 )) s"
