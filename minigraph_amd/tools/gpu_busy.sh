#!/bin/bash
# unprofiled GPU busy share of the pipelined bench steps: samples the driver's gpu_busy_percent (sysfs; rocm-smi as a fallback) every 20 ms while `bench.py --steps N` runs
out=gpurun_out/${1:-busy}.txt
python bench.py --steps ${STEPS:-20} --warmup 2 --one-placement --no-cpu --resident-steps 0 --no-asm --no-small --no-file-out --no-rank-share > gpurun_out/${1:-busy}.json 2> gpurun_out/${1:-busy}.err &
pid=$!
f=$(ls /sys/class/drm/card*/device/gpu_busy_percent 2>/dev/null | head -1)
: > $out
while kill -0 $pid 2>/dev/null; do
	if [ -n "$f" ]; then echo "$(date +%s.%N) $(cat $f 2>/dev/null)" >> $out; else echo "$(date +%s.%N) $(rocm-smi --showuse 2>/dev/null | grep -o 'GPU use (%): [0-9]*' | head -1 | grep -o '[0-9]*$')" >> $out; fi
	sleep 0.02
done
python - "$out" "gpurun_out/${1:-busy}.json" <<'PY'
import sys, json
rows = [l.split() for l in open(sys.argv[1]) if len(l.split()) == 2]
d = json.load(open(sys.argv[2]))
t = [float(r[0]) for r in rows]; b = [float(r[1]) for r in rows]
# the timed steps are the LAST steps * ms_per_step of the run, minus the tail in which bench.py compares and prints (take the window that ends 1 s before the last sample)
dur = d["steps"] * d["ms_per_step"] / 1e3
end = t[-1] - 1.0
win = [x for tt, x in zip(t, b) if end - dur <= tt <= end]
print("samples %d (every %.0f ms), window of the %d timed steps: %d samples, gpu_busy_percent mean %.1f  median %.0f  share of samples at >= 90 %%: %.2f ; value %.3f Gbp/s, %.1f ms per step"
      % (len(b), 1e3 * (t[-1] - t[0]) / max(1, len(t) - 1), d["steps"], len(win), sum(win) / max(1, len(win)), sorted(win)[len(win) // 2] if win else -1,
         sum(1 for x in win if x >= 90) / max(1, len(win)), d["value"], d["ms_per_step"]))
PY
