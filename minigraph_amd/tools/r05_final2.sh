#!/bin/bash
# second half of the round's evidence: the bench line again, now that the counter / PMC files it prices its kernels with (profiles/r05z_*) are those of this tree; and the
# isolated kernel statistics with graph chaining on the device
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 400 python bench.py --steps 20 --warmup 5 --workdir /tmp/mga_wd > $out/r05z_bench_steps20.json 2> $out/r05z_bench_steps20.err
echo "[bench] rc $? $(( $(date +%s) - t0 )) s"
PROF_WORKDIR=/tmp/mga_wd PROF_PARTS="dev" timeout 200 bash minigraph_amd/tools/prof_all.sh r05z > $out/r05z_prof_dev.log 2>&1
echo "[prof dev] rc $? $(( $(date +%s) - t0 )) s"
