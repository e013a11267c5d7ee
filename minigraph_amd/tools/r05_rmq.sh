#!/bin/bash
# round 5, GPU call: the RMQ chainer's forward pass on the device -- parity tests, then the 500 Mbp -cx asm job with the passes on the device and on the host threads
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "rmq_forward or asm_preset_long or both_presets or mt_known" 2>&1 | tail -15 | tee $out/r05d_tests_rmq.txt
echo "[tests rmq] $(( $(date +%s) - t0 )) s"
timeout 600 python -m pytest tests/test_gpu_stages.py -q -x -m gpu -k "wfa" 2>&1 | tail -3 | tee $out/r05d_tests_wfa.txt
echo "[tests wfa] $(( $(date +%s) - t0 )) s"
A="--genome 500000000 --chr 10 --n 10 --contig 50000000 --cigar-only --keep-ref /tmp/asm_wd"
MGA_DEBUG_PIPE=1 python minigraph_amd/tools/asm_check.py $A > $out/r05d_asm_dev.txt 2> $out/r05d_asm_dev.err; tail -2 $out/r05d_asm_dev.txt; grep "\[rq\]" $out/r05d_asm_dev.err | tail -4
echo "[asm dev + ref] $(( $(date +%s) - t0 )) s"
MGA_DEBUG_PIPE=1 MGA_DEV_RMQ=0 python minigraph_amd/tools/asm_check.py $A > $out/r05d_asm_host.txt 2> $out/r05d_asm_host.err; tail -1 $out/r05d_asm_host.txt; grep "\[rq\]" $out/r05d_asm_host.err | tail -4
python minigraph_amd/tools/asm_check.py $A > $out/r05d_asm_dev2.txt 2> /dev/null; tail -1 $out/r05d_asm_dev2.txt
echo "[asm] $(( $(date +%s) - t0 )) s"
