#!/bin/bash
# graph-chaining placement A/B on one box, alternating (box drift is larger than the effect):  place_ab.sh <rounds>
# each line: the headline step of bench.py with all chunks on the host threads (pct 0), a quarter / half on the device, all on the device
n=${1:-2}
B="python bench.py --steps ${STEPS:-6} --warmup 1 --one-placement --no-cpu --resident-steps 0 --no-asm --no-small --no-file-out --no-rank-share"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2]); print("%-28s %.3f Gbp/s  %.1f ms/step  cpu %.2f s/step" % (sys.argv[1], d["value"], d["ms_per_step"], d["host"]["cpu_s_per_step"]), flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", sys.argv[2][-200:])
PY
}
for r in $(seq 1 $n); do
	show "host (pct 0)" "$(env MGA_DEV_GCHAIN_PCT=0 $B 2>/dev/null | tail -1)"
	show "device (all)" "$($B --placement device 2>/dev/null | tail -1)"
	show "pct 25" "$(env MGA_DEV_GCHAIN_PCT=25 $B 2>/dev/null | tail -1)"
	show "pct 50" "$(env MGA_DEV_GCHAIN_PCT=50 $B 2>/dev/null | tail -1)"
done
