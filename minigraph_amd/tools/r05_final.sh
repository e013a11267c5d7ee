#!/bin/bash
# the round's evidence set from ONE tree: the bench line (20 steps), kernel stats / PMC / SQ counters of the bench command (tools/prof_all.sh), then the whole GPU test suite
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
tag=${1:-r05z}
t0=$(date +%s)
timeout 400 python bench.py --steps 20 --warmup 5 --workdir /tmp/mga_wd > $out/${tag}_bench_steps20.json 2> $out/${tag}_bench_steps20.err
echo "[bench] rc $? $(( $(date +%s) - t0 )) s"; tail -c 600 $out/${tag}_bench_steps20.json | cut -c1-600
PROF_WORKDIR=/tmp/mga_wd PROF_PARTS="${PROF_PARTS:-iso pipe pmc sq}" PROF_SQ_LIGHT=1 timeout 600 bash minigraph_amd/tools/prof_all.sh $tag > $out/${tag}_prof_all.log 2>&1
echo "[prof] rc $? $(( $(date +%s) - t0 )) s"
timeout 500 python -u -m pytest tests -q -x -m gpu 2>&1 | tail -6 | tee $out/${tag}_gpu_tests.txt
echo "[tests] $(( $(date +%s) - t0 )) s"
