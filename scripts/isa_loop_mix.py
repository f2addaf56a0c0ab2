#!/usr/bin/env python
"""Instruction mix of the largest loop of a kernel in a gfx950 .s file.  usage: isa_loop_mix.py file.s <substring of kernel symbol>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
start = None
for m in re.finditer(r'^(_Z\w+):', s, re.M):
    if pat in m.group(1):
        start = m
        break
body = s[start.end():]
body = body[:body.index('s_endpgm')]
lines = body.split('\n')
labels = {}
for i, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = i
loops = []
for i, l in enumerate(lines):
    mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        loops.append((labels[mm.group(1)], i))


def mix(a, b):
    ins = [l.strip().split()[0] for l in lines[a:b] if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    g = collections.Counter()
    for k, v in c.items():
        if k.startswith('v_mfma'): g['mfma'] += v
        elif k.startswith('v_'): g['valu'] += v
        elif k.startswith('ds_'): g['lds'] += v
        elif k.startswith('s_waitcnt'): g['waitcnt'] += v
        elif k.startswith('s_'): g['salu'] += v
        elif k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): g['vmem'] += v
    return len(ins), dict(g), c


print("whole kernel:", mix(0, len(lines))[:2])
print("loops (start,end,len):", [(a, b, b - a) for a, b in loops])
if loops:
    a, b = max(loops, key=lambda t: t[1] - t[0])
    n, g, c = mix(a, b)
    print("largest loop:", n, g)
    print(c.most_common(45))
