#!/usr/bin/env python
"""tools/wi_sections.py engine.s: static instruction counts of wiener_istft_kernel<true> between labels and barriers (what the frame loop issues)."""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
i0 = next(i for i, l in enumerate(lines) if l.startswith('_ZN3umx19wiener_istft_kernelILb1EE'))
i1 = next(i for i in range(i0, len(lines)) if 's_endpgm' in lines[i])
cnt = [Counter(label='entry')]
for l in lines[i0 + 1:i1]:
    t = l.strip()
    if re.match(r'\.LBB\d+_\d+:', t):
        cnt.append(Counter(label=t)); continue
    if not t or t[0] in ';.': continue
    op = t.split()[0]
    if op == 's_barrier': cnt.append(Counter(label='barrier'))
    k = 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else 'salu' if op.startswith('s_') else 'vmem' if op.startswith(('buffer', 'global', 'flat', 'scratch')) else 'other'
    cnt[-1][k] += 1
    if op.startswith('scratch'): cnt[-1]['scratch'] += 1
for c in cnt:
    print({k: c[k] for k in ('label', 'valu', 'salu', 'lds', 'vmem', 'scratch') if c[k]})
