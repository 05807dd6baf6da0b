#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr text on stdin or a file)."""
import re, subprocess, sys
rows = []; cur = {}
for l in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    m = re.search(r'remark: (Function Name): (\S+)', l) or re.search(r'remark:\s+(.*?): (.*?) \[', l)
    if not m:
        if ' error' in l: print(l.rstrip())
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()}; rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    print('%-96s V=%-4s S=%-4s scratch=%-4s occ=%s lds=%s' % (r['name'][:96], r.get('VGPRs'), r.get('TotalSGPRs'),
          r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))
