timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg; echo "ramp $1 early $2"; MGA_RAMP=$1 MGA_EARLY_RELEASE=$2 timeout 300 python bench.py --no-cpu --steps 3 --warmup 1 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host']['cpu_s_per_step'])"; done
