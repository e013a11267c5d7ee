#!/bin/bash
# FETCH / WRITE passes and the SQ counter pass again on the round's last tree (the files record the hash of the WFA sources they were taken from; bench.py labels others STALE)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
PROF_WORKDIR=/tmp/mga_wd PROF_PARTS="pmc sq" PROF_SQ_LIGHT=1 timeout 200 bash minigraph_amd/tools/prof_all.sh r05z > $out/r05z_prof_pmc_sq.log 2>&1
echo "[prof] rc $? $(( $(date +%s) - t0 )) s"
head -3 $out/r05z_sq_counters.txt | cut -c1-200; python -c "import json; print(json.load(open('$out/r05z_pmc.json'))['wfa_src_sha1'])"
