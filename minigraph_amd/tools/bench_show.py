import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels_ms_isolated"]
print(d["value"], d["ms_per_step"], (d.get("device_placement") or {}).get("value"), d.get("resident"))
print({x:k[x] for x in k if k[x]>0.05})
print({a:b["valu_busy"] for a,b in (d["roofline"].get("valu_busy") or {}).items()})
print(d["parity"])
