#!/bin/bash
# bench.py under a list of environment settings (pipeline knobs), one line per setting:  knob_sweep.sh "MGA_PIPE=5" "MGA_PIPE=6 MGA_WFA_SLOTS=3" ...
# ("-" = no setting; RESIDENT=1 adds the isolated pass and prints the WFA family's kernel times).  Headline placement only, no CPU baseline, no isolated passes: about 40 s per setting on the GPU box.
ulimit -c 0
for s in "$@"; do
	[ "$s" = "-" ] && s=""
	out=$(env $s python bench.py --steps ${STEPS:-4} --warmup 1 --one-placement --no-cpu --resident-steps ${RESIDENT:-0} ${BENCH_ARGS:-} 2>/dev/null | tail -1)
	python - "$s" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    k = d.get("kernels_ms_isolated") or {}
    wfa = sum(v for n, v in k.items() if n.startswith("k_wfa"))
    print("%-40s %.3f Gbp/s  %.1f ms/step  cpu %.2f s/step%s" % (sys.argv[1] or "(default)", d["value"], d["ms_per_step"], d["host"]["cpu_s_per_step"],
          ("  | isolated: k_lchain %.1f  k_sketch %.1f  k_seed %.1f  k_text %.1f  WFA family %.1f ms  " % (k.get("k_lchain", 0), k.get("k_sketch", 0), k.get("k_seed_count", 0) + k.get("k_seed_fill", 0), k.get("k_text", 0), wfa) + " ".join("%s %.1f" % (n.replace("k_wfa", ""), v) for n, v in k.items() if n.startswith("k_wfa") and v > 0.5)) if k else ""), flush=True)
except Exception as e:
    print("%-40s FAILED %s" % (sys.argv[1], sys.argv[2][-200:]))
PY
done
