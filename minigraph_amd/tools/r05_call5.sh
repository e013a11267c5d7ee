#!/bin/bash
# round 5, GPU call 5: full GPU suite of the tree (device RMQ, device target splicing, reachable-only traceback rows, equal chunks), bench A/B of the splicing, asm timing
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -15 | tee $out/r05e_gpu_tests.txt
echo "[tests] $(( $(date +%s) - t0 )) s"
WD=/tmp/mga_wd
python bench.py --steps 3 --warmup 1 --workdir $WD --no-cpu --resident-steps 0 --one-placement --no-asm --no-rank-share > $out/r05e_first.json 2> $out/r05e_first.err; tail -3 $out/r05e_first.err
export STEPS=8 BENCH_ARGS="--workdir $WD --no-asm --no-rank-share"
bash minigraph_amd/tools/knob_sweep.sh - "MGA_DEV_SPLICE=0" - "MGA_DEV_SPLICE=0" "MGA_FRONT_SLOTS=2 MGA_PIPE=6" "MGA_FRONT_SLOTS=2 MGA_PIPE=6 MGA_DEV_SPLICE=0" - 2>&1 | tee $out/r05e_splice_sweep.txt
echo "[sweep] $(( $(date +%s) - t0 )) s"
A="--genome 500000000 --chr 10 --n 10 --contig 50000000 --cigar-only --keep-ref /tmp/asm_wd"
MGA_DEBUG_PIPE=1 python minigraph_amd/tools/asm_check.py $A > $out/r05e_asm_dev.txt 2> $out/r05e_asm_dev.err; tail -1 $out/r05e_asm_dev.txt; grep "\[rq\]" $out/r05e_asm_dev.err | tail -40
python minigraph_amd/tools/asm_check.py $A --no-ref > $out/r05e_asm_dev2.txt 2> /dev/null; tail -1 $out/r05e_asm_dev2.txt
echo "[asm] $(( $(date +%s) - t0 )) s"
