#!/bin/bash
# kernel statistics (isolated, pipelined + overlap summary, device chaining) again on the round's last tree
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
PROF_WORKDIR=/tmp/mga_wd PROF_PARTS="iso pipe dev" timeout 150 bash minigraph_amd/tools/prof_all.sh r05z > $out/r05z_prof_stats.log 2>&1
echo "[prof] rc $? $(( $(date +%s) - t0 )) s"
head -4 $out/r05z_pipe_overlap.txt | cut -c1-200
