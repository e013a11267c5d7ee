#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 45 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "long_read_boundary" 2>&1 | tail -5
