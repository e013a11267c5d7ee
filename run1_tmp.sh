timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k edge_case 2>&1 | grep -v "^$" | tail -15
