import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(d['value'], {k:(v.get('value'), v.get('ms_per_step'), v.get('roofline',{}).get('kernel_ms_per_step'), v.get('roofline',{}).get('host_issue_ms_per_step'), v.get('roofline',{}).get('device_ms_per_step')) for k,v in d['families'].items()})

for k,v in d['families'].items(): print(k, v.get('kernel_breakdown'))
