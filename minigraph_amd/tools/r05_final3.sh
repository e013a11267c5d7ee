#!/bin/bash
# the whole GPU suite on the round's last tree (after r05z: host error paths, k_rmq_fwd's work loop, the scatter kernel's optional stable order) + smoke()
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 330 python -u -m pytest tests -q -x -m gpu 2>&1 | tail -6 | tee $out/r05zz_gpu_tests.txt
echo "[tests] $(( $(date +%s) - t0 )) s"
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $out/r05zz_gpu_tests.txt
