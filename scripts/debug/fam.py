"""Print the family sub-records of a `bench.py` line (value, ms per step, kernel / host-issue / device ms, top kernels): `python scripts/debug/fam.py line.json`.
Used with `family_ctx.py` to find the per-loop prefetch stream that slowed the HSTU / eSASRec legs of the default run."""
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(d['value'], {k:(v.get('value'), v.get('ms_per_step'), v.get('roofline',{}).get('kernel_ms_per_step'), v.get('roofline',{}).get('host_issue_ms_per_step'), v.get('roofline',{}).get('device_ms_per_step')) for k,v in d['families'].items()})

for k,v in d['families'].items(): print(k, v.get('kernel_breakdown'))
