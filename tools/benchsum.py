#!/usr/bin/env python3
"""print the kernel table and the main figures of bench.py output lines: tools/benchsum.py file..."""
import json
import sys
for f in sys.argv[1:]:
    for line in open(f):
        if line.startswith('{'):
            d = json.loads(line)
            print(f, d['config']['name'], d['value'], 'MB/s', d['ms_per_step'], 'ms', d['phases_ms'])
            for k in d['kernels']:
                print('   %-50s %8.3f ms  x%-5s frac %.3f' % (k['kernel'], k['ms_per_step'], k['launches_per_step'], k['frac']))
            print('  ', d['merge_stats'], d['sa_rounds'])
            print('  ', 'digests', d['verified']['outputs_match_reference_digests_whole_text'], 'host', d['host_buffer_boundary'], 'cli', d.get('cli_file_to_file'))
            print('  ', {k: (v['frac'] if v else None) for k, v in d['roofline_passes'].items()})
