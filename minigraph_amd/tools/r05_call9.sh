#!/bin/bash
# round 5, GPU call 9: two reads per wavefront in k_lchain's first-pass DP: parity, then isolated kernel time and step time with and without
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_stages.py -q -x -m gpu -k "lchain" 2>&1 | tail -8 | tee $out/r05i_tests_lchain.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "mt_known or synthetic_vs_reference or parity_sweep or long_join or command_line" 2>&1 | tail -8 | tee $out/r05i_tests_e2e.txt
echo "[tests] $(( $(date +%s) - t0 )) s"
WD=/tmp/mga_wd
python bench.py --steps 3 --warmup 1 --workdir $WD --no-cpu --resident-steps 0 --one-placement --no-asm --no-rank-share > /dev/null 2>&1
export STEPS=8 RESIDENT=1 BENCH_ARGS="--workdir $WD --no-asm --no-rank-share"
bash minigraph_amd/tools/knob_sweep.sh - "MGA_LC_PAIR=0" - "MGA_LC_PAIR=0" 2>&1 | tee $out/r05i_lcpair_sweep.txt
echo "[sweep] $(( $(date +%s) - t0 )) s"
