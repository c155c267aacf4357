#!/usr/bin/env python3
"""Static instruction census of one kernel in hipcc -S output: VALU / SALU / LDS / VMEM per region, regions split at
s_barrier.  usage: isa_count.py file.s mangled-substring"""
import re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(_Z\S*%s\S*):[^\n]*\n" % re.escape(key), s, re.M)
i = m.end()
j = s.index('.end_amdhsa_kernel', i) if '.end_amdhsa_kernel' in s[i:] else len(s)
j = s.index('s_endpgm', i)
region, counts = 0, [dict(valu=0, salu=0, lds=0, vmem=0, other=0)]
for l in s[i:j].split('\n'):
    l = l.strip()
    if not l or l.startswith(('.', ';', '_')) or l.endswith(':'):
        continue
    op = l.split()[0]
    if op == 's_barrier':
        counts.append(dict(valu=0, salu=0, lds=0, vmem=0, other=0)); continue
    c = counts[-1]
    if op.startswith('v_'): c['valu'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['vmem'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    else: c['other'] += 1
print(m.group(1))
for k, c in enumerate(counts):
    print('region', k, c)
