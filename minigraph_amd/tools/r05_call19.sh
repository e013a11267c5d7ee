#!/bin/bash
# issue-rate microbenchmark with the packed instructions of the WFA step and with partly empty EXEC masks (why do the WFA rungs issue MORE vector instructions per second than
# the full-EXEC v_max stream that prices roofline.valu_busy?)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 minigraph_amd/tools/valu_rate.hip -o /tmp/valu_rate 2> $out/r05q_valu_build.err
timeout 120 /tmp/valu_rate > $out/r05q_valu_rate_exec.txt 2>&1
cat $out/r05q_valu_rate_exec.txt | cut -c1-230
