#!/bin/bash
# round 5, GPU call 1: (A) bench line of the tree, (B) timestamped kernel trace of the PIPELINED step + overlap analysis, (C) the host-side stage timeline,
# (D) the VALU-rate microbenchmark alone and under the SQ counters (pins the counters' units), (E) pipeline knob sweep, (F) the tests the round touched so far.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
WD=/tmp/mga_wd
COMMON="--workdir $WD --no-cpu --resident-steps 0 --one-placement --no-asm --no-rank-share"
t0=$(date +%s)
python bench.py --steps 6 --warmup 2 $COMMON > $out/r05a_base.json 2> $out/r05a_base.err
echo "[A] base done $(( $(date +%s) - t0 )) s"; tail -c 300 $out/r05a_base.err
# (B)
rm -rf $out/prof_r05a_pipe; mkdir -p $out/prof_r05a_pipe
rocprofv3 --kernel-trace --output-format csv -d $out/prof_r05a_pipe -- python bench.py --steps 3 --warmup 1 $COMMON > $out/r05a_pipe_bench_under_rocprof.json 2> $out/r05a_pipe_rocprof.err
T=$(find $out/prof_r05a_pipe -name "*kernel_trace.csv" | head -1)
if [ -n "$T" ]; then
	python minigraph_amd/tools/trace_overlap.py "$T" --window 0.5,1.0 --title "pipelined bench step (host placement, 16 threads), steps after the warm-up" > $out/r05a_pipe_overlap.txt 2>&1
	python minigraph_amd/tools/trace_overlap.py "$T" --title "whole run (index build + warm-up + steps)" > $out/r05a_pipe_overlap_all.txt 2>&1
	gzip -9 < "$T" > $out/r05a_pipe_kernel_trace.csv.gz
	head -60 $out/r05a_pipe_overlap.txt
fi
rm -rf $out/prof_r05a_pipe
echo "[B] trace done $(( $(date +%s) - t0 )) s"
# (C)
MGA_DEBUG_PIPE=2 python bench.py --steps 1 --warmup 1 $COMMON > /dev/null 2> $out/r05a_pipe_timeline.txt
echo "[C] timeline done $(( $(date +%s) - t0 )) s"
# (D)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 minigraph_amd/tools/valu_rate.hip -o /tmp/valu_rate 2> $out/r05a_valu_build.err
/tmp/valu_rate > $out/r05a_valu_rate.txt 2>&1
rm -rf $out/prof_r05a_valu
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/prof_r05a_valu -- /tmp/valu_rate > $out/r05a_valu_rate_under_pmc.txt 2> $out/r05a_valu_pmc.err
C=$(find $out/prof_r05a_valu -name "*counter_collection.csv" | head -1)
[ -n "$C" ] && cp "$C" $out/r05a_valu_counters.csv && python minigraph_amd/tools/prof_summary.py --calib "$C" > $out/r05a_valu_calibration.txt 2>&1
rm -rf $out/prof_r05a_valu
cat $out/r05a_valu_calibration.txt | head -40
echo "[D] calibration done $(( $(date +%s) - t0 )) s"
# (E)
STEPS=6 BENCH_ARGS="--workdir $WD --no-asm --no-rank-share" bash minigraph_amd/tools/knob_sweep.sh - "MGA_CUT=0" "MGA_TAIL=2" "MGA_TAIL=3" "MGA_WFA_GRID_PCT=75" "MGA_WFA_GRID_PCT=50" \
	"MGA_WFA_SLOTS=3" "MGA_WFA_SLOTS=3 MGA_WFA_GRID_PCT=50" "MGA_WFA_SLOTS=1" "MGA_TAIL=3 MGA_WFA_GRID_PCT=75" "MGA_DEV_GCHAIN=1" "MGA_DEV_GCHAIN=1 MGA_TAIL=3" - 2>&1 | tee $out/r05a_knob_sweep.txt
echo "[E] sweep done $(( $(date +%s) - t0 )) s"
# (F)
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "knobs or launches_its_own_ranks or chunks_beyond or mt_known or one_input_four" 2>&1 | tail -5 | tee $out/r05a_tests.txt
echo "[F] tests done $(( $(date +%s) - t0 )) s"
