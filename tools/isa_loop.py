#!/usr/bin/env python3
"""Print the instruction mix of one kernel from a -save-temps .s file: tools/isa_loop.py file.s <substring> [dump]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
starts = [m for m in re.finditer(r'^(_Z\w+):', s, re.M) if pat in m.group(1)]
for m in starts:
    i0 = m.end()
    i1 = s.index('.Lfunc_end', i0)
    lines = []
    for l in s[i0:i1].split('\n'):
        l = l.split(';')[0].strip() if not l.strip().startswith('.LBB') else l.strip().split(';')[0].strip()
        if l and (not l.startswith('.') or l.startswith('.LBB')):
            lines.append(l)
    print(m.group(1)[:60], len(lines), 'lines')
    lab = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(':')}
    # loops = backward branches
    for i, l in enumerate(lines):
        mm = re.match(r's_c?branch\w*\s+(\.LBB\w+)', l)
        if mm and mm.group(1) in lab and lab[mm.group(1)] < i:
            body = lines[lab[mm.group(1)]:i + 1]
            c = collections.Counter()
            for b in body:
                op = b.split()[0]
                if op.endswith(':'): continue
                k = ('valu64' if re.match(r'v_\w*f64|v_rsq_f64|v_rndne_f64|v_ldexp_f64', op) else 'valu32' if op.startswith('v_') else
                     'ds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else
                     'salu' if op.startswith('s_') else 'other')
                c[k] += 1
            print(f'  loop {mm.group(1)} [{lab[mm.group(1)]}..{i}] {dict(c)}')
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write('\n'.join(lines))
