#!/bin/bash
# round 5, GPU call 2: does leaving wave slots / LDS to the other chunks' kernels buy overlap?  (persistent WFA grids at a fraction of the chip, one / two / three chunks in the
# WFA phase, two chunks in the front phase, more pipeline threads) + the walk of a rung next to the next rung's forward pass (MGA_WFA_TB_SIDE)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
WD=/tmp/mga_wd
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_stages.py -q -x -m gpu -k "wfa" 2>&1 | tail -3 | tee $out/r05c_tests_wfa.txt
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "knobs or mt_known or synthetic_vs_reference or parity_sweep" 2>&1 | tail -3 | tee $out/r05c_tests_e2e.txt
echo "[tests] $(( $(date +%s) - t0 )) s"
python bench.py --steps 3 --warmup 1 --workdir $WD --no-cpu --resident-steps 0 --one-placement --no-asm --no-rank-share > $out/r05c_first.json 2> $out/r05c_first.err; tail -5 $out/r05c_first.err
export STEPS=8 BENCH_ARGS="--workdir $WD --no-asm --no-rank-share"
bash minigraph_amd/tools/knob_sweep.sh - "MGA_WFA_TB_SIDE=0" "MGA_WFA_GRID_PCT=50" "MGA_WFA_GRID_PCT=50 MGA_WFA_SLOTS=1" "MGA_WFA_GRID_PCT=50 MGA_WFA_SLOTS=1 MGA_FRONT_SLOTS=2" \
	"MGA_WFA_GRID_PCT=50 MGA_FRONT_SLOTS=2 MGA_PIPE=6" "MGA_WFA_GRID_PCT=35 MGA_FRONT_SLOTS=2 MGA_PIPE=6" "MGA_FRONT_SLOTS=2 MGA_PIPE=6" \
	"MGA_WFA_GRID_PCT=62 MGA_WFA_SLOTS=1 MGA_FRONT_SLOTS=2 MGA_PIPE=6" "MGA_WFA_GRID_PCT=50 MGA_WFA_SLOTS=1 MGA_FRONT_SLOTS=2 MGA_PIPE=6" \
	"MGA_WFA_GRID_PCT=50 MGA_WFA_SLOTS=3 MGA_FRONT_SLOTS=2 MGA_PIPE=8" "MGA_WFA_GRID_PCT=75 MGA_WFA_SLOTS=3 MGA_FRONT_SLOTS=3 MGA_PIPE=8" - "MGA_WFA_TB_SIDE=0" \
	"MGA_WFA_GRID_PCT=50 MGA_WFA_SLOTS=1 MGA_FRONT_SLOTS=2 MGA_PIPE=6" "MGA_TAIL=3" "MGA_TAIL=3 MGA_WFA_GRID_PCT=50 MGA_WFA_SLOTS=1 MGA_FRONT_SLOTS=2 MGA_PIPE=6" - 2>&1 | tee $out/r05c_knob_sweep.txt
echo "[sweep host placement] $(( $(date +%s) - t0 )) s"
BENCH_ARGS="$BENCH_ARGS --placement device" bash minigraph_amd/tools/knob_sweep.sh - "MGA_WFA_GRID_PCT=50 MGA_WFA_SLOTS=1 MGA_FRONT_SLOTS=2 MGA_PIPE=6" "MGA_WFA_GRID_PCT=50 MGA_FRONT_SLOTS=2 MGA_PIPE=6" "MGA_FRONT_SLOTS=2 MGA_PIPE=6" - 2>&1 | tee $out/r05c_knob_sweep_device.txt
echo "[sweep device placement] $(( $(date +%s) - t0 )) s"
