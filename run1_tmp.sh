timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
MGA_DEBUG_PIPE=1 timeout 300 python bench.py --reads 100000 --steps 2 --warmup 1 2>&1 | grep "host CPU\|metric\|rror" | tail -3 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['host'], d.get('parity'), d['kernels_ms'].get('k_text'))
    else: print(l.rstrip()[:400])
"
