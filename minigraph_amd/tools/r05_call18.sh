#!/bin/bash
# after the k_rmq_fwd work-loop restructure and the mapper's error paths: the -x asm tests, the device RMQ test, the e2e subset
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 280 python -u -m pytest tests/test_gpu_e2e.py -v -x -m gpu -k "rmq or asm or mt_known or synthetic_vs_reference or knobs" 2>&1 | grep -v "^$" | tee $out/r05p_tests_asm.txt | tail -25
echo "[tests] rc ${PIPESTATUS[0]} $(( $(date +%s) - t0 )) s"
