#!/bin/bash
# bench.py under a list of environment settings (pipeline knobs), one line per setting:  knob_sweep.sh "MGA_PIPE=5" "MGA_PIPE=6 MGA_WFA_SLOTS=3" ...
# ("-" = no setting).  Headline placement only, no CPU baseline, no isolated passes: about 40 s per setting on the GPU box.
ulimit -c 0
for s in "$@"; do
	[ "$s" = "-" ] && s=""
	out=$(env $s python bench.py --steps ${STEPS:-4} --warmup 1 --one-placement --no-cpu --resident-steps 0 ${BENCH_ARGS:-} 2>/dev/null | tail -1)
	python - "$s" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    print("%-40s %.3f Gbp/s  %.1f ms/step  cpu %.2f s/step" % (sys.argv[1] or "(default)", d["value"], d["ms_per_step"], d["host"]["cpu_s_per_step"]), flush=True)
except Exception as e:
    print("%-40s FAILED %s" % (sys.argv[1], sys.argv[2][-200:]))
PY
done
