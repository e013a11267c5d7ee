#!/bin/bash
# last sanity of the library as committed: golden / reference-binary e2e subset + smoke()
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 70 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "mt_known or synthetic_vs_reference" 2>&1 | tail -3
timeout 30 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
