#!/usr/bin/env python3
"""Order of the memory / MFMA / barrier instructions of one kernel in a device assembly file (hipcc -S
--cuda-device-only): M mfma, L global load, S global store, W ds_write, R ds_read, B barrier, w s_waitcnt.
usage: isa_shape.py file.s <mangled-name-substring> [max chars]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'\n(\S*' + re.escape(pat) + r'\S*):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S):
    ins = [l.strip().split()[0] for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith((';', '.', '//'))]
    seq = []
    for i in ins:
        k = ('M' if 'mfma' in i else 'L' if i.startswith(('buffer_load', 'global_load')) else 'S' if i.startswith(('buffer_store', 'global_store'))
             else 'W' if i.startswith('ds_write') else 'R' if i.startswith('ds_read') else 'B' if 's_barrier' in i
             else 'w' if i.startswith('s_waitcnt') else 'j' if i.startswith(('s_cbranch', 's_branch')) else None)
        if k:
            seq.append(k)
    t = ''.join(seq)
    out = re.sub(r'(.)\1*', lambda g: g.group(1) + (str(len(g.group(0))) if len(g.group(0)) > 1 else ''), t)
    print(m.group(1)[:90], len(ins), 'instructions')
    print(out[: int(sys.argv[3]) if len(sys.argv) > 3 else 3000])
