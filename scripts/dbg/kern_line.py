import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d.get('kernels_ms_per_step') or {}
print(d['ms_per_step'], {n:round(v,4) for n,v in k.items() if 'sdf' in n or 'color' in n or 'wgrad' in n})
