#!/bin/bash
# packed rungs (MGA_WFA_PACKED=1): stage tests under a SHORT timeout, and only when they pass the e2e subset and the sweep
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 170 python -m pytest tests/test_gpu_stages.py -q -x -m gpu -k "wfa" 2>&1 | tail -15 | tee $out/r05n_tests_wfa_packed.txt
rc=${PIPESTATUS[0]}
echo "[tests] rc $rc $(( $(date +%s) - t0 )) s"
[ $rc -ne 0 ] && exit 0
timeout 120 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "mt_known or synthetic_vs_reference" 2>&1 | tail -8 | tee $out/r05n_tests_e2e_packed.txt
rc=${PIPESTATUS[0]}
echo "[tests e2e] rc $rc $(( $(date +%s) - t0 )) s"
[ $rc -ne 0 ] && exit 0
WD=/tmp/mga_wd
timeout 60 python bench.py --steps 3 --warmup 1 --workdir $WD --no-cpu --resident-steps 0 --one-placement --no-asm --no-rank-share > /dev/null 2>&1
STEPS=6 RESIDENT=1 BENCH_ARGS="--workdir $WD --no-asm --no-rank-share" timeout 150 bash minigraph_amd/tools/knob_sweep.sh - "MGA_WFA_PACKED=7" "MGA_WFA_PACKED=0" 2>&1 | tee $out/r05n_packed_sweep.txt
echo "[sweep] $(( $(date +%s) - t0 )) s"
