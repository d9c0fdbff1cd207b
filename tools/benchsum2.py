"""print the interesting parts of a bench.py JSON line: python tools/benchsum2.py file.json"""
import json
import sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value', r['value'], 'ms_per_step', r['ms_per_step'], 'phases', r['phases_ms'])
for k, v in (r.get('roofline_passes') or {}).items():
    if v:
        print('  %-40s %8.3f ms  frac %.4f' % (k[:40], v['ms'], v['frac']))
print('verified', r.get('verified'))
print('mem', r.get('device_memory'))
for k in r['kernels']:
    print('  %-50s %8.3f ms %6.1f launches  frac %.4f' % (k['kernel'], k['ms_per_step'], k['launches_per_step'], k['frac']))
if r.get('north_star'):
    print('north_star', json.dumps(r['north_star'])[:1500])
if r.get('rccl'):
    print('rccl', r['rccl'])
print('cpu', r.get('cpu_baseline'))
